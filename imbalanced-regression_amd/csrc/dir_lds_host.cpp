// K8 — LDS per-sample weights on the host (once per run; a bit-exactness problem, not a
// throughput problem — SURVEY.md §8 a11). Replaces imdb-wiki-dir/datasets.py:55-83.
// Compiled with -ffp-contract=off and without fast-math: every operation is one IEEE rounding,
// in the order numpy 2.2 / scipy 1.15 perform them.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>
#include "dir_hip.h"

namespace {

// numpy's float32 pairwise summation (numpy/_core/src/umath/loops_utils.h.src, PW_BLOCKSIZE 128).
float pairwise_sum_f32(const float* a, int64_t n) {
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
}

// np.sum over a contiguous float32 array (numpy 2.2.6): the reduction iterator hands the add loop
// chunks of the ufunc buffer size (np.getbufsize() = 8192 elements); each chunk is pairwise-summed
// and the chunk sums are accumulated left to right. Verified bit for bit against np.sum for
// n in 1..300, 8191..8194, 12208, 16384/5, 20000, 100003, 191509 (build container).
float numpy_sum_f32(const float* a, int64_t n) {
    const int64_t kBuf = 8192;
    float res = pairwise_sum_f32(a, n < kBuf ? n : kBuf);
    for (int64_t i = kBuf; i < n; i += kBuf) res += pairwise_sum_f32(a + i, (n - i) < kBuf ? (n - i) : kBuf);
    return res;
}

// scipy.ndimage.convolve1d(x, w, mode='constant', cval=0) for odd ks, float64 arithmetic.
// Symmetric windows take scipy's pairwise branch: centre tap first, then (x[c+j] + x[c-j]) * w
// from the outermost pair inwards (ni_filters.c NI_Correlate1D); others the plain branch with
// the window reversed (convolution == correlation with the flipped kernel).
void convolve1d_constant(const double* x, int n, const double* w, int ks, double* out) {
    const int h = ks / 2;
    std::vector<double> pad(n + 2 * h, 0.0);
    for (int i = 0; i < n; ++i) pad[i + h] = x[i];
    bool symmetric = true, antisymmetric = true;
    for (int k = 1; k <= h; ++k) {
        if (std::fabs(w[h + k] - w[h - k]) > DBL_EPSILON) symmetric = false;
        if (std::fabs(w[h + k] + w[h - k]) > DBL_EPSILON) antisymmetric = false;
    }
    std::vector<double> fw(w, w + ks);                  // convolve1d flips the weights
    for (int k = 0; k < ks / 2; ++k) std::swap(fw[k], fw[ks - 1 - k]);
    for (int i = 0; i < n; ++i) {
        const double* c = pad.data() + i + h;           // centre
        double t;
        if (symmetric) {
            t = c[0] * fw[h];
            for (int jj = -h; jj < 0; ++jj) t += (c[jj] + c[-jj]) * fw[jj + h];
        } else if (antisymmetric) {
            t = c[0] * fw[h];
            for (int jj = -h; jj < 0; ++jj) t += (c[jj] - c[-jj]) * fw[jj + h];
        } else {
            t = c[h] * fw[2 * h];                       // scipy starts from the last tap
            for (int jj = -h; jj < h; ++jj) t += c[jj] * fw[jj + h];
        }
        out[i] = t;
    }
}

}  // namespace

extern "C" int dir_lds_weights(const double* labels, int64_t n, int max_target, int reweight, int lds,
                               const double* window, int ks, float* weights) {
    if (!labels || !weights || n <= 0 || max_target <= 0) return DIR_EINVAL;
    if (reweight != DIR_REWEIGHT_SQRT_INV && reweight != DIR_REWEIGHT_INVERSE) return DIR_EINVAL;
    if (lds && (!window || ks <= 0 || (ks & 1) == 0)) return DIR_EINVAL;
    std::vector<int32_t> bin(n);
    std::vector<int64_t> counts(max_target, 0);
    for (int64_t i = 0; i < n; ++i) {
        const double l = labels[i];
        if (!(l >= 0.0)) return DIR_EINVAL;             // negative / NaN label: the reference raises KeyError
        int64_t b = (int64_t)l;                         // int(label): truncation (datasets.py:63)
        if (b > max_target - 1) b = max_target - 1;
        bin[i] = (int32_t)b;
        counts[b] += 1;
    }
    std::vector<double> value(max_target);
    const bool integer_valued = (reweight == DIR_REWEIGHT_INVERSE);
    for (int k = 0; k < max_target; ++k) {
        if (reweight == DIR_REWEIGHT_SQRT_INV) value[k] = std::sqrt((double)counts[k]);          // :64-65
        else value[k] = (double)(counts[k] < 5 ? 5 : (counts[k] > 1000 ? 1000 : counts[k]));       // :66-67
    }
    if (lds) {                                                                                   // :73-78
        std::vector<double> sm(max_target);
        convolve1d_constant(value.data(), max_target, window, ks, sm.data());
        for (int k = 0; k < max_target; ++k)
            value[k] = integer_valued ? (double)(int64_t)sm[k] : sm[k];    // int64 output: C truncation (A.5)
    }
    for (int64_t i = 0; i < n; ++i) weights[i] = (float)(1.0 / value[bin[i]]);                   // :80
    const float total = numpy_sum_f32(weights, n);                                            // :81
    const float scaling = (float)n / total;
    for (int64_t i = 0; i < n; ++i) weights[i] = scaling * weights[i];                           // :82
    return DIR_OK;
}

extern "C" int dir_abi_version(void) { return DIR_ABI_VERSION; }

extern "C" const char* dir_error_string(int code) {
    switch (code) {
    case DIR_OK: return "ok";
    case DIR_EINVAL: return "invalid argument";
    case DIR_EUNSUPPORTED: return "unsupported dtype/mode in this build";
    case DIR_EWORKSPACE: return "workspace too small";
    default: return code > 0 ? "HIP runtime error (code is a hipError_t)" : "unknown error";
    }
}
