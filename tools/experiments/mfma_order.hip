// build: hipcc -O2 -ffp-contract=off --offload-arch=gfx950 mfma_order.hip -o mfma_order ; result on MI355X: "fma-chain 1024" of 1024
// Does v_mfma_f32_32x32x2_f32 accumulate like a k-ordered chain of fused multiply-adds?
// One wave: C[32][32] = sum_k A[32][k] B[k][32], k < KD, by MFMA (2 k per instruction) and by three scalar candidates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KD = 768;
__global__ void k(const float* A, const float* B, float* out) {   // A [32][KD], B [32][KD] (row = output col)
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < KD; k0 += 2) {
        const float a = A[r * KD + k0 + h];
        const float b = B[r * KD + k0 + h];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out[row * 32 + r] = acc[i];
    }
}
int main() {
    std::vector<float> A(32 * KD), B(32 * KD), C(1024);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int eq_chain = 0, eq_pair = 0, eq_mul = 0, eq_pairfirst = 0;
    for (int row = 0; row < 32; ++row)
        for (int col = 0; col < 32; ++col) {
            float c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f;
            for (int kk = 0; kk < KD; ++kk) c1 = fmaf(A[row * KD + kk], B[col * KD + kk], c1);          // fma chain
            for (int kk = 0; kk < KD; kk += 2) {                                                          // exact pair sum, one rounding
                double p = (double)A[row * KD + kk] * B[col * KD + kk] + (double)A[row * KD + kk + 1] * B[col * KD + kk + 1];
                c2 = (float)((double)c2 + p);
            }
            for (int kk = 0; kk < KD; ++kk) c3 = c3 + A[row * KD + kk] * B[col * KD + kk];              // mul then add
            for (int kk = 0; kk < KD; kk += 2) {                                                          // fma(a1,b1, a0*b0) then add
                float p = fmaf(A[row * KD + kk + 1], B[col * KD + kk + 1], A[row * KD + kk] * B[col * KD + kk]);
                c4 = c4 + p;
            }
            const float m = C[row * 32 + col];
            eq_chain += (m == c1); eq_pair += (m == c2); eq_mul += (m == c3); eq_pairfirst += (m == c4);
        }
    printf("of 1024 outputs: fma-chain %d, exact-pair %d, mul-add %d, pair-first %d\n", eq_chain, eq_pair, eq_mul, eq_pairfirst);
    return 0;
}
