// yh_hyp.cpp — yh_hyp_test: the binomial presence test of `yacht run` (R3) in host C++, no scipy.
//
// Restates single_hyp_test / get_alt_mut_rate (src/yacht/hypothesis_recovery_src.py:209-306): for every organism
//     p     = ani_thresh ** ksize
//     n     = int(n_excl * min_coverage)                       (double product, truncated)
//     thr   = binom.ppf(1 - significance, n, p)                smallest k with cdf(k) >= q
//     conf  = 1 - binom.cdf(thr, n, p)
//     alt   = 1 - (1 - betaincinv(n - thr, 1 + thr, significance)) ** (1 / ksize),  NaN -> -1
//     p_val = binom.cdf(n_match, n, p) if n_match <= n else 1
//     in_sample_est = (n_match >= thr) and (n_match != 0)
// The reference gets these from scipy (Boost.Math underneath).  Here: binomial point probabilities by Loader's
// saddle-point form (C. Loader, "Fast and accurate computation of binomial probabilities", 2000: log pmf from
// Stirling-series errors and the deviance terms, relative error ~1e-15 -- lgamma differences lose 5+ digits at
// n ~ 1e5), the distribution function as the SHORTER tail summed outward from k by the exact term ratio in log
// scale (no underflow before the result itself underflows), the quantile by walking that function from the
// normal-approximation guess, and the inverse regularized incomplete beta -- for the integer arguments this path
// only ever has, I_x(n - t, t + 1) = P[Bin(n, 1 - x) <= t] -- by safeguarded Newton on the same function.
// thr / conf / alt depend on n alone (p, significance, ksize are per call): computed once per distinct n.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/yacht_hip.h"

void yh_set_error(const char* fmt, ...);

namespace {

// The point probability is evaluated in long double (x87 80-bit on the hosts this runs on): its logarithm reaches
// ~-700 before the result underflows, and a double there carries an absolute error of ~5e-13 -- the size of the
// disagreement between this file and scipy on p-values around 1e-250 before the change (both were that far from the
// exact value, on opposite sides).
typedef long double ld;

// log(n!) - log(sqrt(2 pi n) (n/e)^n) for integer n >= 1 (0 for n = 0: never used as a factor)
ld stirlerr(ld n) {
    static const std::vector<ld> small = [] {
        std::vector<ld> t(16, 0.0L);
        ld fact = 1.0L;
        for (int i = 1; i <= 15; ++i) {
            fact *= (ld)i;  // exact up to 15!
            t[i] = logl(fact) - ((ld)i + 0.5L) * logl((ld)i) + (ld)i - 0.918938533204672741780329736406L;
        }
        return t;
    }();
    if (n <= 15.0L) return small[(int)n];
    const ld S0 = 1.0L / 12.0L, S1 = 1.0L / 360.0L, S2 = 1.0L / 1260.0L, S3 = 1.0L / 1680.0L, S4 = 1.0L / 1188.0L,
             S5 = 691.0L / 360360.0L;
    const ld nn = n * n;
    if (n > 500) return (S0 - (S1 - S2 / nn) / nn) / n;
    if (n > 80) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
    return (S0 - (S1 - (S2 - (S3 - (S4 - S5 / nn) / nn) / nn) / nn) / nn) / n;
}

// x log(x / np) + np - x without cancellation near x = np
ld bd0(ld x, ld np) {
    if (fabsl(x - np) < 0.1L * (x + np)) {
        ld v = (x - np) / (x + np);
        ld s = (x - np) * v;
        ld ej = 2 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; ++j) {
            ej *= v;
            const ld s1 = s + ej / ((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
        return s;
    }
    return x * logl(x / np) + np - x;
}

// log P[Bin(n, p) = x], 0 <= x <= n, q = 1 - p
ld log_pmf(double xd, double nd, double pd, double qd) {
    const ld x = xd, n = nd, p = pd, q = qd;
    if (p <= 0.0L) return x == 0 ? 0.0L : -INFINITY;
    if (q <= 0.0L) return x == n ? 0.0L : -INFINITY;
    if (x == 0) return n * (p < 0.5L ? log1pl(-p) : logl(q));  // (whichever of p, q is the small, exactly known one)
    if (x == n) return n * (q < 0.5L ? log1pl(-q) : logl(p));
    const ld lc = stirlerr(n) - stirlerr(x) - stirlerr(n - x) - bd0(x, n * p) - bd0(n - x, n * q);
    const ld lf = 1.837877066409345483560659472811L + logl(x) + log1pl(-x / n);
    return lc - 0.5L * lf;
}

// P[Bin(n, p) <= k]; q = 1 - p is passed in: the caller knows which of the two is exact
double binom_cdf(double k, double n, double p, double q) {
    if (k < 0) return 0.0;
    if (k >= n) return 1.0;
    if (p <= 0.0) return 1.0;
    if (q <= 0.0) return 0.0;  // (k < n)
    if ((k + 1.0) <= (n + 1.0) * p) {  // k below the mode: the lower tail, terms falling from i = k down
        const ld l0 = log_pmf(k, n, p, q);
        ld t = 1.0L, s = 1.0L;
        for (double i = k; i > 0; i -= 1.0) {
            t *= ((ld)i * q) / ((ld)(n - i + 1.0) * p);
            s += t;
            if (t < s * 1e-22L) break;
        }
        return (double)expl(l0 + logl(s));
    }
    // k + 1 at or above the mode: the upper tail, terms falling from i = k + 1 up
    const ld l0 = log_pmf(k + 1.0, n, p, q);
    ld t = 1.0L, s = 1.0L;
    for (double i = k + 1.0; i < n; i += 1.0) {
        t *= ((ld)(n - i) * p) / ((ld)(i + 1.0) * q);
        s += t;
        if (t < s * 1e-22L) break;
    }
    return (double)(1.0L - expl(l0 + logl(s)));
}

// smallest k in [0, n] with cdf(k) >= prob
double binom_ppf(double prob, double n, double p) {
    if (n <= 0) return 0.0;
    if (!(prob > 0.0)) return -1.0;  // (scipy: a - 1)
    if (prob >= 1.0) return n;
    const double q = 1.0 - p;
    // normal approximation of the quantile as the starting point
    double z = 0.0;
    {   // Acklam's rational approximation of the normal quantile (a starting guess only)
        const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
        const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
        const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
        const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
        if (prob < 0.02425) {
            const double u = sqrt(-2 * log(prob));
            z = (((((c[0] * u + c[1]) * u + c[2]) * u + c[3]) * u + c[4]) * u + c[5]) / ((((d[0] * u + d[1]) * u + d[2]) * u + d[3]) * u + 1);
        } else if (prob > 1 - 0.02425) {
            const double u = sqrt(-2 * log1p(-prob));
            z = -(((((c[0] * u + c[1]) * u + c[2]) * u + c[3]) * u + c[4]) * u + c[5]) / ((((d[0] * u + d[1]) * u + d[2]) * u + d[3]) * u + 1);
        } else {
            const double u = prob - 0.5, r = u * u;
            z = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * u / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
        }
    }
    double k = floor(n * p + z * sqrt(n * p * q) + 0.5);
    k = std::min(std::max(k, 0.0), n);
    if (binom_cdf(k, n, p, q) >= prob) {
        while (k > 0 && binom_cdf(k - 1.0, n, p, q) >= prob) k -= 1.0;
    } else {
        do k += 1.0; while (k < n && binom_cdf(k, n, p, q) < prob);
    }
    return k;
}

// x with I_x(a, b) = y for a = n - t > 0, b = t + 1 (integers): I_x(a, b) = P[Bin(n, 1 - x) <= t]; NaN otherwise
double betaincinv_int(double n, double t, double y) {
    const double a = n - t;
    if (!(a > 0.0) || !(t >= 0.0) || !(y >= 0.0) || !(y <= 1.0)) return NAN;
    if (y == 0.0) return 0.0;
    if (y == 1.0) return 1.0;
    auto g = [&](double x) { return binom_cdf(t, n, 1.0 - x, x); };
    // derivative: x^(a-1) (1-x)^(b-1) / B(a, b) = n * P[Bin(n - 1, x) = a - 1]
    auto dg = [&](double x) { return n * (double)expl(log_pmf(a - 1.0, n - 1.0, x, 1.0 - x)); };
    double lo = 0.0, hi = 1.0;
    double x = a / (a + t + 1.0);  // the mean of Beta(a, b)
    x = std::min(std::max(x, 1e-300), 1.0 - 1e-16);
    for (int it = 0; it < 300; ++it) {
        const double gx = g(x);
        if (gx == y) return x;
        if (gx < y) lo = x; else hi = x;
        const double d = dg(x);
        double xn = (d > 0 && std::isfinite(d)) ? x - (gx - y) / d : NAN;
        if (!(xn > lo && xn < hi)) xn = (lo > 0 && hi / lo > 4.0) ? sqrt(lo * hi) : 0.5 * (lo + hi);  // bisection (geometric across decades)
        if (xn == x || fabs(xn - x) <= 2.2e-16 * fabs(x) * 0.5) { x = xn; break; }
        x = xn;
        if (hi - lo <= 1.1e-16 * hi) break;
    }
    return x;
}

struct PerN {
    double thr, conf, alt;
};

}  // namespace

extern "C" int yh_hyp_test(uint64_t n, const uint32_t* n_excl, const uint32_t* n_match, int ksize, double significance,
                           double ani_thresh, double min_coverage, uint8_t* in_sample_est, double* p_val,
                           uint32_t* n_excl_cov, double* threshold, double* confidence, double* alt_mut_rate) {
    if (n && (!n_excl || !n_match || !in_sample_est || !p_val || !n_excl_cov || !threshold || !confidence || !alt_mut_rate)) {
        yh_set_error("yh_hyp_test: null argument");
        return YH_ERR_INVALID_ARG;
    }
    if (ksize < 1 || !(significance >= 0.0 && significance <= 1.0) || !(ani_thresh >= 0.0 && ani_thresh <= 1.0) ||
        !(min_coverage >= 0.0 && min_coverage <= 1.0)) {
        yh_set_error("yh_hyp_test: ksize >= 1, and significance, ani_thresh, min_coverage in [0, 1]");
        return YH_ERR_INVALID_ARG;
    }
    const double p = pow(ani_thresh, (double)ksize);
    const double q_prob = 1.0 - significance;
    std::vector<uint32_t> ncov(n);
    for (uint64_t i = 0; i < n; ++i) {
        ncov[i] = (uint32_t)((double)n_excl[i] * min_coverage);  // int(x * cov): truncation of the double product
        n_excl_cov[i] = ncov[i];
    }
    std::vector<uint32_t> uniq(ncov);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    std::vector<PerN> per(uniq.size());
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, 32), (n + 255) / 256));
    auto for_range = [&](uint64_t total, auto&& body) {
        if (T <= 1 || total < 64) { body(0, total); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) {
            // interleaved blocks: costs grow with n, and `uniq` is sorted
            th.emplace_back([&, t] { for (uint64_t i = t; i < total; i += T) body(i, i + 1); });
        }
        for (auto& x : th) x.join();
    };
    for_range(uniq.size(), [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            const double nn = (double)uniq[i];
            PerN r;
            r.thr = binom_ppf(q_prob, nn, p);
            r.conf = 1.0 - binom_cdf(r.thr, nn, p, 1.0 - p);
            const double x = betaincinv_int(nn, r.thr, significance);
            const double mut = 1.0 - pow(1.0 - x, 1.0 / (double)ksize);
            r.alt = std::isnan(mut) ? -1.0 : mut;
            per[i] = r;
        }
    });
    for_range(n, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            const PerN& r = per[std::lower_bound(uniq.begin(), uniq.end(), ncov[i]) - uniq.begin()];
            threshold[i] = r.thr;
            confidence[i] = r.conf;
            alt_mut_rate[i] = r.alt;
            const double m = (double)n_match[i];
            p_val[i] = (n_match[i] <= ncov[i]) ? binom_cdf(m, (double)ncov[i], p, 1.0 - p) : 1.0;
            in_sample_est[i] = (m >= r.thr && n_match[i] != 0) ? 1 : 0;
        }
    });
    return YH_OK;
}
