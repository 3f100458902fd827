// One-sided register Jacobi (gingr_amd/csrc/eig.hip sym_eig_cols_kernel): time and accuracy on Gram matrices of a truncated
// pivoted-Cholesky-like factor, n = 34 (rank-100 model), 100, 128, 171 (rank-512 model), 192.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I gingr_amd/csrc -I include tools/ubench_sym_eig.hip -o tools/bin/ubench_sym_eig
#include "../gingr_amd/csrc/eig.hip"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
int main() {
    gingr_ctx *ctx = nullptr;
    if (gingr_ctx_create(0, &ctx)) return 1;
    std::mt19937_64 rng(3);
    std::normal_distribution<double> nd;
    for (int n : {2, 5, 34, 64, 65, 100, 128, 129, 171, 192}) {
        const int m = n + 40;
        std::vector<double> L((size_t)m * n), G((size_t)n * n);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) L[(size_t)i * n + j] = nd(rng) * std::pow(0.97, j);  // spectrum over ~ 2.5 decades at n = 192
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double s = 0;
                for (int i = 0; i < m; ++i) s += L[(size_t)i * n + a] * L[(size_t)i * n + b];
                G[(size_t)a * n + b] = s;
            }
        double *dG, *dW, *dE, *dV;
        int32_t *dI;
        hipMalloc(&dG, (size_t)n * n * 8);
        hipMalloc(&dW, sym_eig_cols_work_doubles(n) * 8);
        hipMalloc(&dE, n * 8);
        hipMalloc(&dV, (size_t)n * n * 8);
        hipMalloc(&dI, 8);
        hipMemcpy(dG, G.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
        const double *Gs[1] = {dG};
        double *Ws[1] = {dW}, *Es[1] = {dE}, *Vs[1] = {dV};
        int32_t *Is[1] = {dI};
        const int32_t ld[1] = {n}, ns[1] = {n};
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, ctx->stream);
            for (int i = 0; i < 5; ++i) launch_sym_eig_cols(ctx, 1, Gs, ld, ns, Ws, Es, Vs, Is);
            hipEventRecord(b, ctx->stream);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        std::vector<double> ev(n), V((size_t)n * n);
        int32_t info[2];
        hipMemcpy(ev.data(), dE, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(V.data(), dV, (size_t)n * n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(info, dI, 8, hipMemcpyDeviceToHost);
        double res = 0, orth = 0, tr = 0, trg = 0;
        bool desc = true;
        for (int k = 0; k < n; ++k) {
            tr += ev[k];
            trg += G[(size_t)k * n + k];
            if (k && ev[k] > ev[k - 1]) desc = false;
            double rk = 0;
            for (int i = 0; i < n; ++i) {
                double s = 0;
                for (int j = 0; j < n; ++j) s += G[(size_t)i * n + j] * V[(size_t)j * n + k];
                s -= ev[k] * V[(size_t)i * n + k];
                rk += s * s;
            }
            res = std::fmax(res, std::sqrt(rk) / ev[k]);
            for (int l = 0; l < n; ++l) {
                double s = 0;
                for (int i = 0; i < n; ++i) s += V[(size_t)i * n + k] * V[(size_t)i * n + l];
                orth = std::fmax(orth, std::fabs(s - (k == l ? 1.0 : 0.0)));
            }
        }
        printf("n %3d  %8.1f us  sweeps %d singular %d  cond %.3g  max |G v - l v| / l %.2e  max |V^T V - I| %.2e  trace rel %.2e  descending %d\n", n,
               ms * 1e3 / 5, info[0], info[1], ev[0] / ev[n - 1], res, orth, std::fabs(tr - trg) / trg, (int)desc);
        hipFree(dG); hipFree(dW); hipFree(dE); hipFree(dV); hipFree(dI);
    }
    return 0;
}
