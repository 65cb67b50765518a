"""Reduce the two PMC passes of tools/profile_round.sh (pmc_fetch.txt / pmc_write.txt, KB per dispatch per kernel) to the
HBM-side bytes per launch of the dominant kernel class (all gemm_nt2_kernel epilogue variants, launch-weighted) and write
profiles/rNN_pmc_traffic.json, which bench.py reports as roofline.traffic.
FETCH_SIZE is doubled: on gfx950 it tallies the 128-B requests of 16-B-per-lane coalesced reads at 64 B
(MI355X_MICROARCH.md, HBM section)."""
import json, re, sys

def load(path):
    rows = []
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows.append((m.group(1).strip(), int(m.group(3)), float(m.group(4))))
    return rows

def main(fetch, write, out, tag):
    def weighted(rows):
        sel = [(n, d, v) for n, d, v in rows if "gemm_nt2_kernel" in n and not re.search(r"2, 3, 0, 2, 64>|ELi3ELi0E", n)]
        n = sum(d for _, d, _ in sel)
        return n, sum(d * v for _, d, v in sel) / n
    nf, f = weighted(load(fetch))
    nw, w = weighted(load(write))
    # algorithmic bytes per launch, launch-weighted over the 8 NT GEMMs of a layer at config 2 (T = 19712, W = 768):
    T, W = 19712, 768
    def gemm(M, N, K, out_b, extra=0):
        return M * K * 2 + N * K * 2 + M * N * out_b + extra
    # round 4: the out-projection and c_proj store bf16 (the residual add moved into the following LayerNorm); only the last
    # block's c_proj keeps the fp32 residual epilogue.  fc writes u + act'(pre) (2 x bf16), d-activation reads act'(pre).
    layers = 12
    per_layer = [gemm(T, 3 * W, W, 2), gemm(T, W, W, 2), gemm(T, 4 * W, W, 4), gemm(T, 4 * W, W, 2, T * 4 * W * 2),
                 gemm(T, W, 4 * W, 2), gemm(T, W, W, 2), gemm(T, W, 3 * W, 2)]
    cproj_bf16 = gemm(T, W, 4 * W, 2)
    # round 5 (pooled last block): the last block keeps only its qkv projection and that projection's data gradient on all T
    # rows; its other six products run on the B pooled rows through the 128 x 128 kernel and are not in this kernel class
    algo = ((layers - 1) * (sum(per_layer) + cproj_bf16) + per_layer[0] + per_layer[6]) / (8 * (layers - 1) + 2)
    j = {"kernel": "gemm_nt2_kernel (all epilogue variants, launch-weighted)",
         "launches_fetch_pass": nf, "launches_write_pass": nw,
         "fetch_size_kb_raw_per_launch": round(f, 1), "write_size_kb_per_launch": round(w, 1),
         "fetch_correction": "x2 (MI355X_MICROARCH.md, HBM section: gfx950 FETCH_SIZE tallies 128-B requests at 64 B for 16 B/lane coalesced reads)",
         "traffic_bytes_per_launch": int(round((2 * f + w) * 1024)),
         "algorithmic_bytes_per_launch": int(round(algo)),
         "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 3 --warmup 1 "
                   f"--no-cpu-baseline --no-packed --no-prof`, tools/profile_round.sh {tag}; reduced by tools/pmc_traffic.py"}
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:5])
