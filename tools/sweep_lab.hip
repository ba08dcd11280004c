// sweep_lab.hip - ablation timings of k_sweep_erm (not part of the library; built and run by
// hand:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I../admm-for-rank-based-loss_amd/csrc
//        tools/sweep_lab.hip -o gpurun_out/sweep_lab).  Prints ms and GB/s per variant.
#include "../admm-for-rank-based-loss_amd/csrc/sweep_erm.hip"
#include <cstdio>
#include <vector>

// stubs for the symbols sweep_erm.hip expects from the rest of the library
int reduce_blocks() { return 1024; }
int launch_sum_partials(const double*, int, int, double*, hipStream_t) { return 0; }
int launch_loss_sum(int, int64_t, const double*, double, double*, double*, hipStream_t) { return 0; }
void rbl_set_error(const char*, ...) {}

template <int P, int R, int S, bool WL, int EXP, bool ONE = false>
static void run(const char* name, const float* D, long long n, long long ld, double* w, double* z, double* lam, double* v,
                double* zn, double* pred, double* slab, double* partials, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, tot = 0.f;
    const int reps = 6;
    for (int i = 0; i < reps + 1; ++i) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_sweep_erm<float, 0, P, R, S, WL, EXP, ONE>), dim3(grid), dim3(SE_THREADS), 0, 0, D, n, ld, w, z, lam, v, zn,
                           1.0, 1e-3, pred, slab, partials);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i) { tot += ms; best = ms < best ? ms : best; }
    }
    printf("%-34s P=%d R=%d S=%d WL=%d ONE=%d EXP=%2d grid=%4d  avg %.3f ms  best %.3f ms  %.0f GB/s\n", name, P, R, S, (int)WL, (int)ONE, EXP, grid,
           tot / reps, best, (double)n * ld * 4 / (tot / reps) * 1e-6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 6000000, ld = 1000;
    float* D;
    double *w, *z, *lam, *v, *zn, *pred, *slab, *partials;
    hipMalloc(&D, n * ld * 4);
    hipMemsetD32((hipDeviceptr_t)D, 0x3c23d70a /* 0.01f */, n * ld);
    hipMalloc(&w, ld * 8); hipMalloc(&z, n * 8); hipMalloc(&lam, n * 8); hipMalloc(&v, n * 8); hipMalloc(&zn, n * 8);
    hipMalloc(&pred, 16); hipMalloc(&slab, 4096 * ld * 8); hipMalloc(&partials, 4096 * 3 * 8);
    std::vector<double> hw(ld, 1e-3), hz(n, 0.05);
    double hp[2] = {1.02e-3, 0.0};
    hipMemcpy(w, hw.data(), ld * 8, hipMemcpyHostToDevice);
    hipMemcpy(z, hz.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(lam, hz.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(pred, hp, 16, hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cu = prop.multiProcessorCount;
#define RUN(name, P, R, S, WL, EXP, grid) run<P, R, S, WL, EXP>(name, D, n, ld, w, z, lam, v, zn, pred, slab, partials, grid)
#define RUN1(name, P, R, S, WL, EXP, grid) run<P, R, S, WL, EXP, true>(name, D, n, ld, w, z, lam, v, zn, pred, slab, partials, grid)
    RUN1("full (library)", 4, 2, 8, false, 0, 2 * cu);
    RUN("full, two-copy loop", 4, 2, 8, false, 0, 2 * cu);
    RUN1("full, 1 block/CU", 4, 2, 8, false, 0, cu);
    RUN1("full, R=4 S=4 w in LDS", 4, 4, 4, true, 0, 2 * cu);
    RUN1("v-only (rank-weighted problems)", 4, 2, 8, false, SE_VONLY, 2 * cu);
    RUN1("v-only, no row writes", 4, 2, 8, false, SE_VONLY | 1, 2 * cu);
    RUN1("v-only, writes into an L2 ring", 4, 2, 8, false, SE_VONLY | 256, 2 * cu);
    RUN1("v-only, nt stores", 4, 2, 8, false, SE_VONLY | 512, 2 * cu);
    RUN1("v-only, plain stores (round 2)", 4, 2, 8, false, SE_VONLY | 4096, 2 * cu);
    RUN1("v-only, buffer store aux=0", 4, 2, 8, false, SE_VONLY | 2048 | (0 << 13), 2 * cu);
    RUN1("v-only, buffer store sc0", 4, 2, 8, false, SE_VONLY | 2048 | (1 << 13), 2 * cu);
    RUN1("v-only, buffer store nt", 4, 2, 8, false, SE_VONLY | 2048 | (2 << 13), 2 * cu);
    RUN1("v-only, buffer store sc1", 4, 2, 8, false, SE_VONLY | 2048 | (16 << 13), 2 * cu);
    RUN1("v-only, buffer store sc0 sc1", 4, 2, 8, false, SE_VONLY | 2048 | (17 << 13), 2 * cu);
    RUN1("v-only, buffer store sc1 nt", 4, 2, 8, false, SE_VONLY | 2048 | (18 << 13), 2 * cu);
    RUN1("v-only, buffer store sc0 sc1 nt", 4, 2, 8, false, SE_VONLY | 2048 | (19 << 13), 2 * cu);
    RUN1("v-only, S*R = 64 rows", 4, 4, 16, false, SE_VONLY, 2 * cu);
    RUN1("v-only, 1 block/CU", 4, 2, 8, false, SE_VONLY, cu);
    RUN1("v-only, R=1 S=16", 4, 1, 16, false, SE_VONLY, 2 * cu);
    RUN1("full, writes into an L2 ring", 4, 2, 8, false, 256, 2 * cu);
    RUN1("full, nt stores", 4, 2, 8, false, 512, 2 * cu);
    RUN1("full, plain stores (round 2)", 4, 2, 8, false, 4096, 2 * cu);
    RUN1("v-only, no wave reduce", 4, 2, 8, false, SE_VONLY | 8, 2 * cu);
    RUN1("v-only, no side loads", 4, 2, 8, false, SE_VONLY | 32, 2 * cu);
    RUN1("v-only, no writes/reduce/side", 4, 2, 8, false, SE_VONLY | 1 | 8 | 32, 2 * cu);
    RUN1("q-only (D^T c)", 4, 2, 8, false, SE_QONLY, 2 * cu);
    RUN1("q-only R=4 S=4", 4, 4, 4, false, SE_QONLY, 2 * cu);
    RUN1("q-only, 1 block/CU", 4, 2, 8, false, SE_QONLY, cu);
    RUN1("q-only, 3 blocks/CU", 4, 2, 8, false, SE_QONLY, 3 * cu);
    {
        double* vsave = v;
        v = nullptr;
        RUN1("v-only, lambda store only (v NULL)", 4, 2, 8, false, SE_VONLY, 2 * cu);
        RUN1("full, lambda + z' stores only (v NULL)", 4, 2, 8, false, 0, 2 * cu);
        v = vsave;
    }
    RUN1("no row writes", 4, 2, 8, false, 1, 2 * cu);
    RUN1("no acc phase", 4, 2, 8, false, 2, 2 * cu);
    RUN1("no prox", 4, 2, 8, false, 4, 2 * cu);
    RUN1("no wave reduce", 4, 2, 8, false, 8, 2 * cu);
    RUN1("no dot", 4, 2, 8, false, 16, 2 * cu);
    RUN1("loads only", 4, 2, 8, false, 1 | 2 | 4 | 8 | 16 | 32, 2 * cu);
    RUN1("loads only, 4 blocks/CU", 4, 2, 8, false, 1 | 2 | 4 | 8 | 16 | 32, 4 * cu);
    return 0;
}
