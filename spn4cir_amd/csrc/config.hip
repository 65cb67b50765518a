// Run-time configuration of the library: every SPN_* environment variable is captured ONCE, when the library is loaded;
// the A/B switches of the kernels read that snapshot (spn_env), never the live environment, and spn_config_dump() reports
// it - bench.py prints it in its JSON line, so a stray variable that changes a kernel is visible next to the number.
#include "common.h"
#include "../../include/spn4cir_hip.h"
#include <string.h>
#include <string>
#include <utility>
#include <vector>

extern char** environ;

namespace {
struct EnvSnapshot {
    std::vector<std::pair<std::string, std::string>> kv;
    EnvSnapshot() {
        for (char** e = environ; e && *e; ++e) {
            if (strncmp(*e, "SPN_", 4) != 0) continue;
            const char* eq = strchr(*e, '=');
            if (!eq) continue;
            kv.emplace_back(std::string(*e, eq - *e), std::string(eq + 1));
        }
    }
};
const EnvSnapshot& snapshot() {
    static const EnvSnapshot s;
    return s;
}
__attribute__((constructor)) void spn_capture_env() { (void)snapshot(); }
}  // namespace

const char* spn_env(const char* name) {
    for (const auto& p : snapshot().kv)
        if (p.first == name) return p.second.c_str();
    return nullptr;
}

int spn_stream_wt() {
    static const int v = [] {
        const char* e = spn_env("SPN_STREAM_WT");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return v;
}

extern "C" int spn_config_dump(char* buf, int cap) {
    std::string s = "{\"experiments_build\": ";
#ifdef SPN_EXPERIMENTS
    s += "1";
#else
    s += "0";
#endif
    s += ", \"env\": {";
    bool first = true;
    for (const auto& p : snapshot().kv) {
        if (!first) s += ", ";
        first = false;
        s += "\"";
        for (char c : p.first) if (c != '"' && c != '\\' && (unsigned char)c >= 32) s += c;
        s += "\": \"";
        for (char c : p.second) if (c != '"' && c != '\\' && (unsigned char)c >= 32) s += c;
        s += "\"";
    }
    s += "}}";
    if (buf && cap > 0) {
        const size_t n = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size() + 1;
}
