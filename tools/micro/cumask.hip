// Which compute units does a bit of hipExtStreamCreateWithCUMask stand for on this chip?  A stream is created with the
// first NCLEAR bits of the mask cleared (argv[1], default 16), a kernel of many one-wave workgroups records the XCD and
// the hardware id of every workgroup, and the host prints, per XCD, how many distinct compute units were used.
// Build: hipcc --offload-arch=gfx950 -O2 -o cumask cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

__global__ void k_where(unsigned *out) {
    // a little work so that the workgroups spread over the chip instead of draining through a few units
    double a = threadIdx.x;
    for (int i = 0; i < 20000; i++) a = a * 1.0000001 + 1e-9;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); // XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc + (a < 0 ? 1u : 0u);
    }
}

static void run(hipStream_t st, const char *what) {
    const int nb = 16384;
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 2 * nb);
    hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nb, hipMemcpyDeviceToHost);
    std::set<unsigned> cus[16];
    int per[16] = {0};
    for (int b = 0; b < nb; b++) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; // CU_ID, SH_ID, SE_ID of HW_ID
        cus[xcc].insert((se << 8) | (sh << 4) | cu);
        per[xcc]++;
    }
    printf("%s:", what);
    int total = 0;
    for (int x = 0; x < 8; x++) { printf("  xcd%d %zu CUs / %d wgs", x, cus[x].size(), per[x]); total += (int)cus[x].size(); }
    printf("  | %d CUs\n", total);
    hipFree(d);
}

int main(int argc, char **argv) {
    const int nclear = argc > 1 ? atoi(argv[1]) : 16;
    hipStream_t plain, masked;
    hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    run(plain, "no mask");
    unsigned mask[8];
    for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
    for (int b = 0; b < nclear; b++) mask[b / 32] &= ~(1u << (b % 32));
    if (hipExtStreamCreateWithCUMask(&masked, 8, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
    char what[64];
    snprintf(what, sizeof(what), "first %d bits cleared", nclear);
    run(masked, what);
    for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
    for (int x = 0; x < 8; x++) mask[x] &= ~3u; // bits 32 x and 32 x + 1
    hipStream_t masked2;
    if (hipExtStreamCreateWithCUMask(&masked2, 8, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
    run(masked2, "bits 32x, 32x+1 cleared");
    return 0;
}
