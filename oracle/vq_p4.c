/* CPU oracle (TEST INFRASTRUCTURE ONLY -- never linked into the product library).
 *
 * Plain-C restatement of the nearest-codebook lookup of the reference:
 *   vq_ae/layers/vq.py:121-129   torch.argmin(torch.cdist(flat_input, embed, ndim,
 *                                 compute_mode='donot_use_mm_for_euclid_dist'), dim=1)
 * with ndim == 4 for NCHW inputs, i.e. a p = 4 Minkowski distance.
 *
 * The arithmetic lives in a third-party dependency (PyTorch ATen, pinned torch 1.11.0+cu115
 * by the reference's pyproject.toml:9; 2.10.0+rocm7.0 in this image).  ATen's CPU cdist
 * forward (aten/src/ATen/native/cpu/DistanceOpsKernel.cpp, run_parallel_cdist + the generic-p
 * functor) is, per (row i, code j):
 *     scalar_t agg = 0;
 *     for c in 0..D-1:  agg = agg + std::pow(std::abs(a[c] - b[c]), p);      // powf, fp32
 *     result = std::pow(agg, 1.0 / p);                                        // double pow -> fp32
 * and argmin(dim=1) returns the lowest index among equal minima.
 * Pinned bit-for-bit against torch.cdist/argmin by tests/golden/make_golden.py (fixture
 * tests/golden/vq_*.npz holds torch's own outputs) and tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <float.h>

void vq_p4_argmin_ref(const float* x, const float* e, long N, long K, long D, float p,
                      int64_t* idx, float* best, float* second, int threads)
{
    const double inv_p = 1.0 / (double)p;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (long n = 0; n < N; ++n) {
        const float* a = x + n * D;
        float b1 = INFINITY, b2 = INFINITY;
        int64_t bi = 0;
        for (long k = 0; k < K; ++k) {
            const float* b = e + k * D;
            float agg = 0.0f;
            for (long c = 0; c < D; ++c)
                agg = agg + powf(fabsf(a[c] - b[c]), p);
            const float fin = (float)pow((double)agg, inv_p);
            if (fin < b1) { b2 = b1; b1 = fin; bi = k; }   /* strict <  => lowest index on ties */
            else if (fin < b2) { b2 = fin; }
        }
        idx[n] = bi;
        if (best) best[n] = b1;
        if (second) second[n] = b2;
    }
}

/* Full distance matrix, for pinning against torch.cdist entry by entry. */
void vq_p4_cdist_ref(const float* x, const float* e, long N, long K, long D, float p, float* out,
                     int threads)
{
    const double inv_p = 1.0 / (double)p;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (long n = 0; n < N; ++n)
        for (long k = 0; k < K; ++k) {
            float agg = 0.0f;
            for (long c = 0; c < D; ++c)
                agg = agg + powf(fabsf(x[n * D + c] - e[k * D + c]), p);
            out[n * K + k] = (float)pow((double)agg, inv_p);
        }
}
