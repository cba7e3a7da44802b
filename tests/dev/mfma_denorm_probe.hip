// Dev probe (not product): does v_mfma_f32_16x16x32_f16 take fp16 DENORMAL inputs at face value (no flush)?  B = raw 4-bit codes in the low mantissa bits
// (value q * 2^-24) and in bits 4..7 (q * 2^-20); A = ordinary fp16.  Prints the max error of D against the exact integer dot product, and the same for
// bf16 denormal-free reference (1024 + q offsets).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
__global__ void k(const u4_t* a, const u4_t* b, f4_t* d) {
    const int lane = threadIdx.x;
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a[lane]), __builtin_bit_cast(h8_t, b[lane]), acc, 0, 0, 0);
    d[lane] = acc;
}
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
int main() {
    // A[m][k] (16 x 32), B[k][n] (32 x 16): lane (kq = lane >> 4, r = lane & 15) holds A[r][8 kq .. 8 kq + 7] and B[8 kq .. 8 kq + 7][r]
    static uint16_t A[16][32], B[32][16];
    static int Q[32][16];
    srand(3);
    for (int shift = 0; shift <= 4; shift += 4) {
        for (int m = 0; m < 16; ++m) for (int kk = 0; kk < 32; ++kk) A[m][kk] = f2h(((rand() % 2001) - 1000) / 1000.0f);
        for (int kk = 0; kk < 32; ++kk) for (int n = 0; n < 16; ++n) { Q[kk][n] = rand() & 15; B[kk][n] = (uint16_t)(Q[kk][n] << shift); }
        u4_t ha[64], hb[64];
        for (int lane = 0; lane < 64; ++lane) {
            const int kq = lane >> 4, r = lane & 15;
            uint16_t ta[8], tb[8];
            for (int e = 0; e < 8; ++e) { ta[e] = A[r][8 * kq + e]; tb[e] = B[8 * kq + e][r]; }
            memcpy(&ha[lane], ta, 16); memcpy(&hb[lane], tb, 16);
        }
        u4_t *da, *db; f4_t* dd; f4_t hd[64];
        hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd));
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
        // D layout: lane (rq = lane >> 4, c = lane & 15) holds D[4 rq + i][c]
        double maxerr = 0, maxref = 0;
        const double unit = ldexp(1.0, -24 + shift);
        for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
            const int m = 4 * (lane >> 4) + i, n = lane & 15;
            double ref = 0;
            for (int kk = 0; kk < 32; ++kk) ref += (double)h2f(A[m][kk]) * Q[kk][n];
            const double got = (double)hd[lane][i] / unit;
            if (fabs(got - ref) > maxerr) maxerr = fabs(got - ref);
            if (fabs(ref) > maxref) maxref = fabs(ref);
        }
        printf("codes at bit %d (fp16 denormals, unit 2^%d): max |D / unit - exact| = %.3g  (max |exact| = %.3g)  -> %s\n", shift, -24 + shift, maxerr, maxref,
               maxerr < 1e-3 * maxref ? "denormal inputs are honoured" : "FLUSHED or wrong");
    }
    return 0;
}
