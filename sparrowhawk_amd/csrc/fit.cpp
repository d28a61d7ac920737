// fit.cpp — automatic min-count from the k-mer spectrum (SPEC S6): two-component Poisson
// mixture, EM in binary64 with a fixed iteration count.  Compiled with -ffp-contract=off so the
// arithmetic is the plain IEEE sequence SPEC S6 prescribes.
// Replaces the crate's `preprocess:*:fitting` phase (AssemblyPage.vue:495,539,552;
// docs/src/assembly.md:16); upstream's model is not in the tree.
#include <math.h>
#include <stdint.h>

namespace shk {

static inline double log_pois(int c, double log_mean, double mean, double lgam) {
    return (double)c * log_mean - mean - lgam;
}

// returns true and sets *out when the fit succeeds
bool spectrum_fit(const uint64_t *histo500, uint32_t *out) {
    const int NB = 500, ITERS = 200;
    double total = 0.0, num = 0.0, den = 0.0;
    for (int c = 1; c <= NB; c++) {
        const double h = (double)histo500[c - 1];
        total += h;
        if (c >= 2) { num += h * (double)c; den += h; }
    }
    if (total == 0.0) return false;
    double w = 0.5;
    double lam = den > 0.0 ? num / den : 2.0;
    if (lam < 2.0) lam = 2.0;
    for (int it = 0; it < ITERS; it++) {
        const double log_1w = log(1.0 - w), log_w = log(w), log_lam = log(lam);
        double acc_w = 0.0, acc_n = 0.0, acc_d = 0.0;
        for (int c = 1; c <= NB; c++) {
            const double h = (double)histo500[c - 1];
            if (h == 0.0) continue;
            const double lg = lgamma((double)c + 1.0);
            const double lp_cov = log_pois(c, log_lam, lam, lg);
            const double lp_err = log_pois(c, 0.0, 1.0, lg);
            const double r = 1.0 / (1.0 + exp((log_1w + lp_cov) - (log_w + lp_err)));
            acc_w += h * r;
            acc_n += h * (1.0 - r) * (double)c;
            acc_d += h * (1.0 - r);
        }
        w = acc_w / total;
        if (w < 1e-9) w = 1e-9;
        if (w > 1.0 - 1e-9) w = 1.0 - 1e-9;
        if (acc_d > 0.0) lam = acc_n / acc_d;
        if (lam < 1.000001) lam = 1.000001;
    }
    if (lam < 2.5) return false;
    const double log_1w = log(1.0 - w), log_w = log(w), log_lam = log(lam);
    for (int c = 2; c <= NB; c++) {
        const double lg = lgamma((double)c + 1.0);
        if (log_1w + log_pois(c, log_lam, lam, lg) > log_w + log_pois(c, 0.0, 1.0, lg)) {
            int v = c - 1;
            if (v < 1) v = 1;
            if (v > 30) v = 30;
            *out = (uint32_t)v;
            return true;
        }
    }
    return false;
}

}  // namespace shk
