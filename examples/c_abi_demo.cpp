// Calling the C ABI of libals_hip.so from plain C++ / HIP - no Python, no torch.
//
// One ALS half-step (the user loop of scripts/als.py:414-433) on a small random problem: build CSR ratings
// and a factor matrix on the host, upload, call als_row_solve, and check every solved row and bias against a
// double-precision solve of the same normal equations on the host.  Then als_predict_at on a few pairs.
// Any other host language binds the same entry points the same way (plain pointers and sizes).
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude examples/c_abi_demo.cpp \
//         -Lcollaborative-filtering_amd/csrc -lals_hip -Wl,-rpath,$PWD/collaborative-filtering_amd/csrc -o c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "als_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)

template <class T>
static T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

// dense Cholesky solve in double (the reference's cholesky_solve, scripts/helpers.py:5-20)
static void solve_spd(std::vector<double>& A, std::vector<double>& b, int k) {
    for (int j = 0; j < k; ++j) {
        for (int p = 0; p < j; ++p) A[j * k + j] -= A[j * k + p] * A[j * k + p];
        A[j * k + j] = std::sqrt(A[j * k + j]);
        for (int i = j + 1; i < k; ++i) {
            for (int p = 0; p < j; ++p) A[i * k + j] -= A[i * k + p] * A[j * k + p];
            A[i * k + j] /= A[j * k + j];
        }
    }
    for (int i = 0; i < k; ++i) { for (int p = 0; p < i; ++p) b[i] -= A[i * k + p] * b[p]; b[i] /= A[i * k + i]; }
    for (int i = k - 1; i >= 0; --i) { for (int p = i + 1; p < k; ++p) b[i] -= A[p * k + i] * b[p]; b[i] /= A[i * k + i]; }
}

int main() {
    const int k = 50, m = 300, n = 120;
    const int ld = als_padded_k(k);
    if (als_version() != ALS_HIP_VERSION || ld != 64) { std::fprintf(stderr, "unexpected library\n"); return 2; }
    const float lam = 2.5f, lam_b = 1.5f;
    const double mu = 3.4;
    std::mt19937 rng(7);
    std::normal_distribution<float> gauss(0.f, 0.5f);
    std::uniform_int_distribution<int> len(1, 90), col(0, n - 1), star(1, 10);

    // CSR ratings: per row a sorted set of distinct columns
    std::vector<int64_t> indptr(m + 1, 0);
    std::vector<int32_t> indices;
    std::vector<float> vals;
    for (int u = 0; u < m; ++u) {
        std::vector<char> used(n, 0);
        const int want = len(rng);
        for (int t = 0; t < want; ++t) used[col(rng)] = 1;
        for (int c = 0; c < n; ++c)
            if (used[c]) { indices.push_back(c); vals.push_back(0.5f * star(rng)); }
        indptr[u + 1] = (int64_t)indices.size();
    }
    // item-side factors [n + 1][ld]: padding columns and the extra last row are zero (F_zero_row)
    std::vector<float> F((size_t)(n + 1) * ld, 0.f), b_i(n), b_u(m, 0.1f);
    for (int i = 0; i < n; ++i) { for (int c = 0; c < k; ++c) F[(size_t)i * ld + c] = gauss(rng); b_i[i] = 0.2f * gauss(rng); }
    std::vector<als_task> tasks(m);
    for (int u = 0; u < m; ++u) tasks[u] = als_task{u, 0, -1, 0};      // every row fits one segment

    int64_t* d_ptr = upload(indptr); int32_t* d_idx = upload(indices); float* d_val = upload(vals);
    float* d_F = upload(F); float* d_bi = upload(b_i); float* d_bu = upload(b_u);
    als_task* d_tasks = upload(tasks);
    std::vector<double> mu_h(1, mu); double* d_mu = upload(mu_h);
    std::vector<float> zeros((size_t)(m + 1) * ld, 0.f); float* d_X = upload(zeros);
    std::vector<int32_t> st(1, 0); int32_t* d_status = upload(st);
    if (!d_ptr || !d_idx || !d_val || !d_F || !d_bi || !d_bu || !d_tasks || !d_mu || !d_X || !d_status) return 2;

    // operand scale of the f16x2 Gram: ALS_FSCALE_FLOATS floats, zero before the first use (als_row_solve fills it)
    std::vector<float> fs(ALS_FSCALE_FLOATS, 0.f); float* d_fscale = upload(fs);
    if (!d_fscale) return 2;
    als_row_solve_params p = {};
    p.k = k; p.ld = ld; p.nrows = m; p.F_zero_row = n; p.gram_mode = ALS_GRAM_F16X2; p.F_scale = d_fscale;
    p.indptr = d_ptr; p.indices = d_idx; p.vals = d_val; p.F = d_F;
    p.bias_self = d_bu; p.bias_other = d_bi; p.mu = d_mu;
    p.lambda_scalar = lam; p.lambda_bias_scalar = lam_b;
    p.X_out = d_X; p.bias_out = d_bu; p.status = d_status;
    p.tasks = d_tasks; p.ntasks = m;
    const int rc = als_row_solve(&p, nullptr);                         // default stream
    if (rc != 0) { std::fprintf(stderr, "als_row_solve -> %d\n", rc); return 1; }
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> X((size_t)m * ld), bu_new(m);
    HIP_OK(hipMemcpy(X.data(), d_X, X.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(bu_new.data(), d_bu, m * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(st.data(), d_status, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (st[0] != 0) { std::fprintf(stderr, "row %d not SPD\n", st[0] - 1); return 1; }

    double worst_x = 0.0, worst_b = 0.0;
    for (int u = 0; u < m; ++u) {
        std::vector<double> A((size_t)k * k, 0.0), b(k, 0.0);
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        for (int64_t t = lo; t < hi; ++t) {
            const float* f = &F[(size_t)indices[t] * ld];
            const double r = vals[t] - (mu + 0.1f + b_i[indices[t]]);
            for (int a = 0; a < k; ++a) { b[a] += f[a] * r; for (int c = 0; c <= a; ++c) A[a * k + c] += (double)f[a] * f[c]; }
        }
        for (int a = 0; a < k; ++a) A[a * k + a] += lam + 1e-10;
        solve_spd(A, b, k);
        double num = 0.0, scale = 1e-30;
        for (int a = 0; a < k; ++a) { worst_x = std::fmax(worst_x, std::fabs(X[(size_t)u * ld + a] - b[a])); scale = std::fmax(scale, std::fabs(b[a])); }
        for (int64_t t = lo; t < hi; ++t) {
            double dot = 0.0;
            for (int a = 0; a < k; ++a) dot += F[(size_t)indices[t] * ld + a] * b[a];
            num += vals[t] - dot - mu - b_i[indices[t]];
        }
        worst_b = std::fmax(worst_b, std::fabs(bu_new[u] - num / ((double)(hi - lo) + lam_b + 1e-10)));
        (void)scale;
    }
    std::printf("als_row_solve: %d rows, k = %d: max |x - x_ref| = %.3g, max |b_u - ref| = %.3g\n", m, k, worst_x, worst_b);

    // predictions at a few (user, item) pairs: U . Z + mu + b_u + b_i   (scripts/tune_params.py:165-166)
    std::vector<int32_t> us = {0, 1, 2, 299}, is = {3, 5, 119, 0};
    int32_t* d_us = upload(us); int32_t* d_is = upload(is);
    std::vector<float> out(us.size(), 0.f); float* d_out = upload(out);
    if (als_predict_at(k, ld, (int64_t)us.size(), d_us, d_is, d_X, d_F, d_bu, d_bi, d_mu, d_out, nullptr) != 0) return 1;
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    double worst_p = 0.0;
    for (size_t t = 0; t < us.size(); ++t) {
        double ref = mu + bu_new[us[t]] + b_i[is[t]];
        for (int a = 0; a < k; ++a) ref += (double)X[(size_t)us[t] * ld + a] * F[(size_t)is[t] * ld + a];
        worst_p = std::fmax(worst_p, std::fabs(out[t] - ref));
    }
    std::printf("als_predict_at: max |pred - ref| = %.3g\n", worst_p);
    const bool ok = worst_x < 2e-4 && worst_b < 2e-5 && worst_p < 2e-5;
    std::printf(ok ? "OK\n" : "MISMATCH\n");
    return ok ? 0 : 1;
}
