// Differential fuzz of two host builds of the BOBYQA state machine (tests/native/bq_host.cpp against two versions of
// hpmvs_amd/csrc/bobyqa3.hpp): both minimise the same randomly generated objectives -- smooth, noisy, with plateaus of
// the photometric path's failure value 2.0, with tight and loose bounds -- and every evaluation point, value, result
// code, minimum and evaluation count must be bit-identical.  Used when the state machine's memory traffic is
// restructured (the arithmetic and its order must not change):
//   g++ -O2 tools/bq_diff_fuzz.cpp -ldl -o /tmp/bq_diff_fuzz && /tmp/bq_diff_fuzz old.so new.so [problems]
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double (*obj_fn)(unsigned n, const double* x, double* grad, void* data);
typedef int (*min_fn)(obj_fn, void*, double*, const double*, const double*, double, int, double*, int*, double*, int);

struct Problem {
    int kind;
    double a[8], b[8][3], c[8], q[3], ctr[3], plateau_lo, noise;
};
static uint64_t rng_state;
static double urand() {  // splitmix64 -> [0,1)
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static double objective(unsigned, const double* x, double*, void* data) {
    const Problem& P = *(const Problem*)data;
    double f = 0.0;
    for (int i = 0; i < 3; i++) f += P.q[i] * (x[i] - P.ctr[i]) * (x[i] - P.ctr[i]);
    const int terms = P.kind == 0 ? 0 : (P.kind == 1 ? 3 : 8);
    for (int k = 0; k < terms; k++) f += P.a[k] * sin(P.b[k][0] * x[0] + P.b[k][1] * x[1] + P.b[k][2] * x[2] + P.c[k]);
    if (P.noise > 0.0) {  // deterministic high-frequency "texture noise"
        f += P.noise * sin(1e4 * x[0] + 3e4 * x[1] - 2e4 * x[2]);
    }
    if (P.kind >= 3 && x[0] < P.plateau_lo) return 2.0;           // objective_fn's plateau (PatchOptimizer.cpp:311)
    if (P.kind == 4) f = (float)f;                                  // float32-quantised like the photometric mean
    if (P.kind == 5) return 2.0;                                    // all plateau
    return f;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s old.so new.so [problems]\n", argv[0]); return 2; }
    void* ha = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    void* hb = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!ha || !hb) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    min_fn fa = (min_fn)dlsym(ha, "bq3_minimize"), fb = (min_fn)dlsym(hb, "bq3_minimize");
    int (*ra)(void) = (int (*)(void))dlsym(ha, "bq3_rescue_calls");
    if (!fa || !fb) { fprintf(stderr, "bq3_minimize not found\n"); return 2; }
    const int n = argc > 3 ? atoi(argv[3]) : 20000;
    const int cap = 1100;
    std::vector<double> ta(4 * cap), tb(4 * cap);
    rng_state = 0x48504d56ull;
    long evals = 0, bad = 0;
    int hist[16] = {0};
    for (int p = 0; p < n; p++) {
        Problem P;
        P.kind = (int)(urand() * 6.0);
        for (int k = 0; k < 8; k++) {
            P.a[k] = 0.3 * urand();
            for (int i = 0; i < 3; i++) P.b[k][i] = 20.0 * (urand() - 0.5);
            P.c[k] = 6.28 * urand();
        }
        for (int i = 0; i < 3; i++) { P.q[i] = urand() < 0.2 ? 0.0 : 4.0 * urand(); P.ctr[i] = 2.0 * (urand() - 0.5); }
        P.plateau_lo = -0.3 + 0.6 * urand();
        P.noise = urand() < 0.5 ? 0.0 : 1e-3 * urand();
        double lb[3], ub[3], x0[3];
        for (int i = 0; i < 3; i++) {
            const double w = urand() < 0.3 ? 0.05 : 1.5;
            lb[i] = -w * (0.2 + urand()); ub[i] = w * (0.2 + urand());
            x0[i] = urand() < 0.15 ? lb[i] : lb[i] + (ub[i] - lb[i]) * urand();
        }
        const double xtol = urand() < 0.5 ? 1e-7 : 1e-4;
        const int maxeval = urand() < 0.1 ? 40 : 1000;
        double xa[3], xb[3], ma = 0, mb = 0;
        int na = 0, nb = 0;
        memcpy(xa, x0, 24); memcpy(xb, x0, 24);
        std::fill(ta.begin(), ta.end(), 0.0); std::fill(tb.begin(), tb.end(), 0.0);
        const int rca = fa(objective, &P, xa, lb, ub, xtol, maxeval, &ma, &na, ta.data(), cap);
        const int rcb = fb(objective, &P, xb, lb, ub, xtol, maxeval, &mb, &nb, tb.data(), cap);
        evals += na;
        hist[(rca + 8) & 15]++;
        if (rca != rcb || na != nb || memcmp(&ma, &mb, 8) || memcmp(xa, xb, 24) || memcmp(ta.data(), tb.data(), 32 * (size_t)cap)) {
            if (bad < 5) {
                int first = -1;
                for (int i = 0; i < cap && first < 0; i++) if (memcmp(&ta[4 * i], &tb[4 * i], 32)) first = i;
                fprintf(stderr, "problem %d kind %d: rc %d/%d nevals %d/%d first differing evaluation %d\n", p, P.kind, rca, rcb, na, nb, first);
            }
            bad++;
        }
    }
    printf("%d problems, %ld evaluations, rescue calls (first library) %d, result codes:", n, evals, ra ? ra() : -1);
    for (int i = 0; i < 16; i++) if (hist[i]) printf(" %d:%d", i - 8, hist[i]);
    printf("\nmismatching problems: %ld\n", bad);
    return bad ? 1 : 0;
}
