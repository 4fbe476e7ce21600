// Probes the operand/accumulator lane layout of v_mfma_f64_16x16x4_f64 on the device it runs on.
// For every (la, lb): A has a single 1.0 in lane la, B a single 1.0 in lane lb; records where the 1.0 lands in D.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(int* out)   // out[la*64+lb] = lane*4 + r of the non-zero D element, or -1
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double4_t c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r)
                if (c[r] != 0.0) out[la * 64 + lb] = lane * 4 + r;
        }
}
int main()
{
    int* d; int h[4096];
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0xff, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // k(la)==k(lb) iff hit. print for la in {0,1,16,17,4} and lb in {0,1,16,17,4}
    for (int la : {0, 1, 2, 4, 15, 16, 17, 32, 48, 63}) {
        printf("la=%2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" lb%d->(lane%d,r%d)", lb, h[la * 64 + lb] / 4, h[la * 64 + lb] % 4);
        printf("\n");
    }
    return 0;
}
