// pav_lab.hip - where one tile of k_pav_bottom spends its time (not part of the library; built and run by hand:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DPAV_LAB tools/pav_lab.hip -o gpurun_out/pav_lab
//   gpurun_out/pav_lab <ms.bin> <sigma_a.bin> <sigma_b.bin> <rho> <n>).  Input: sorted m and the two rank-weight vectors of the positions as raw doubles
// (tools/pav_tail_probe.py --dump writes the highest tiles of an EHRM run).  The kernel takes its tiles from the high end
// first, so block 0 is the LAST tile; its wall-clock stamps (100 MHz) are printed per stage.
#include <hip/hip_runtime.h>
__device__ long long pav_lab_stamps[8 * 32];
#include "../admm-for-rank-based-loss_amd/csrc/pav.hip"
#include <cstdio>
#include <vector>

int reduce_blocks() { return 1024; }
int launch_sum_partials(const double*, int, int, double*, hipStream_t) { return 0; }
void rbl_set_error(const char*, ...) {}

static std::vector<double> slurp(const char* path, long long n) {
    std::vector<double> v((size_t)n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), sizeof(double), (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "cannot read %lld doubles from %s\n", n, path);
        exit(1);
    }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 6) return 1;
    const double rho = atof(argv[4]);
    const long long n = atoll(argv[5]);
    std::vector<double> ms = slurp(argv[1], n), sga = slurp(argv[2], n), sg = slurp(argv[3], n);
    double *dms, *dsg, *dsa, *du, *dfp;
    u32* dmc;
    int* dbr;
    (void)hipMalloc(&dms, n * 8); (void)hipMalloc(&dsg, n * 8); (void)hipMalloc(&dsa, n * 8); (void)hipMalloc(&du, n * 8); (void)hipMalloc(&dmc, 4);
    (void)hipMalloc(&dfp, 8 * (2 * (n / PB_TILE + 2))); (void)hipMalloc(&dbr, 4);
    const int one = 1;
    (void)hipMemcpy(dbr, &one, 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dms, ms.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsg, sg.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, sga.data(), n * 8, hipMemcpyHostToDevice);
    const unsigned tiles = (unsigned)((n + PB_TILE - 1) / PB_TILE);
    for (int flags = 0; flags < 6; ++flags) {       // bit 0: one wave per seam at the top levels, bit 1: no sequential stage; 4, 5: the speculating EHRM form
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(dmc, 0, 4);
            if (flags >= 4)
                hipLaunchKernelGGL((k_pav_bottom<0, true>), dim3(tiles), dim3(PV_THREADS), 0, 0, dms, dsa, dsg, (const int*)nullptr, rho, n, du,
                                   dmc, (const double*)nullptr, (const double*)nullptr, flags & 1, -1, -5.0, 1, dfp);
            else
                hipLaunchKernelGGL((k_pav_bottom<0, false>), dim3(tiles), dim3(PV_THREADS), 0, 0, dms, dsa, dsg, (const int*)dbr, rho, n, du,
                                   dmc, (const double*)nullptr, (const double*)nullptr, flags, -1, 0.0, 0, (double*)nullptr);
            (void)hipDeviceSynchronize();
        }
        long long st[8 * 32];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(pav_lab_stamps), sizeof(st));
        u32 mc = 0;
        (void)hipMemcpy(&mc, dmc, 4, hipMemcpyDeviceToHost);
        printf("flags %d (speculating form %d, wave_top %d, no_seq %d): merges in all tiles %u; block 0 = the last tile, us per stage (100 MHz clock):\n", flags, flags >= 4, flags & 1,
               flags >= 4 ? 0 : (flags >> 1) & 1, mc);
        const long long* s = st;
        printf("  level 0 + prefixes %.1f | sequential stage %.1f |", (s[1] - s[0]) / 100.0, (s[2] - s[1]) / 100.0);
        long long prev = s[2];
        for (int lv = 0; lv <= 10; ++lv) {
            const long long t = s[3 + lv];
            if (t <= prev) continue;
            printf(" half=%d %.1f |", 1 << lv, (t - prev) / 100.0);
            prev = t;
        }
        printf(" store %.1f | total %.1f\n", (s[20] - prev) / 100.0, (s[20] - s[0]) / 100.0);
        for (int i = 0; i < 8 * 32; ++i) st[i] = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pav_lab_stamps), st, sizeof(st));
    }
    return 0;
}
