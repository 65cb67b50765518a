R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/blas -o b -- python3 $R/tools/blas_ref.py > $R/gpurun_out/blas.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/blas/*.db | head -1) > /dev/null || true
python3 - <<PY
import sqlite3, glob
db = glob.glob("$R/gpurun_out/blas/*.db")[0]
c = sqlite3.connect(db)
q = """select s.kernel_name, count(*), avg(d.end-d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
for name, n, avg in c.execute(q):
    if "Cijk" in name: print(n, round(avg/1e3,1), name)
PY
grep "torch" $R/gpurun_out/blas.log
rm -rf $R/gpurun_out/blas
