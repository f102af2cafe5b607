// Optional per-kernel-class HIP-event timing (bench.py roofline leg).  Off by default: zero overhead.
#include <set>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace {
struct Rec {
    int cls;
    hipEvent_t a, b;
    double flops, bytes, aux;
};
bool g_on = false;
std::vector<Rec> g_recs;
double g_aux[UV_NCLS];
std::set<std::string> g_syms[UV_NCLS];       // kernel symbols launched per class since the last enable(1) (bench.py checks its PMC file against them)
}  // namespace

void uv_prof_enable(int on) {
    if (on && !g_on)
        for (auto& st : g_syms) st.clear();
    g_on = on != 0;
}
std::string uv_prof_symbols(int cls) {
    std::string out;
    if (cls < 0 || cls >= UV_NCLS) return out;
    for (const std::string& sy : g_syms[cls]) out += (out.empty() ? "" : ";") + sy;
    return out;
}
bool uv_prof_on() { return g_on; }
void uv_prof_begin(int cls, double flops, double bytes, hipStream_t s, const char* sym, double aux) {
    if (!g_on) return;
    if (sym && cls >= 0 && cls < UV_NCLS) g_syms[cls].insert(sym);
    Rec r{cls, nullptr, nullptr, flops, bytes, aux};
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void uv_prof_end(hipStream_t s) {
    if (!g_on || g_recs.empty()) return;
    (void)hipEventRecord(g_recs.back().b, s);
}
int uv_prof_collect(double* ms, long* count, double* flops, double* bytes, int ncls) {
    for (int i = 0; i < ncls; ++i) {
        ms[i] = 0;
        count[i] = 0;
        flops[i] = 0;
        bytes[i] = 0;
    }
    for (double& a : g_aux) a = 0;
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.cls < ncls) {
            ms[r.cls] += t;
            count[r.cls] += 1;
            flops[r.cls] += r.flops;
            bytes[r.cls] += r.bytes;
            if (r.cls >= 0 && r.cls < UV_NCLS) g_aux[r.cls] += r.aux;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    return UV_OK;
}
void uv_prof_aux(double* aux, int ncls) {
    for (int i = 0; i < ncls; ++i) aux[i] = i < UV_NCLS ? g_aux[i] : 0.0;
}
