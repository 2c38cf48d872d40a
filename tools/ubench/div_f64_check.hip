// Development tool: the float64 sigmoid's divide (device_math.h: div_unit_range_f64 -- v_rcp_f64, two Newton steps, quotient,
// remainder, one correction; no v_div_scale / v_div_fmas / v_div_fixup) against the compiler's IEEE `/` on the operands the
// E-step produces: den = 1 + e, num = e or 1, e = exp(-|x|) anywhere in [0, 1] (subnormal e included).  2^32 pairs.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I viprs_amd/csrc tools/ubench/div_f64_check.hip -o build/ubench/div_f64_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#include "device_math.h"

using namespace viprs;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ void check(unsigned long long* mism, unsigned long long* first, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(i), g = mix64(h);
        // e: random mantissa, exponent uniform over [-1080, 0] (every binade of (0, 1], subnormals, a few zeros) -- or, every
        // fourth pair, a value of the form the exp actually returns near 1 (dense there)
        const double m = 1.0 + (double)(h >> 12) * 0x1p-52;
        double e = ldexp(m, -(int)(g % 1081u) - 1);
        if ((i & 3) == 3) e = 1.0 - (double)(h >> 12) * 0x1p-53;
        if (e > 1.0) e = 1.0;
        const double den = 1.0 + e;
        const double num = (g >> 40) & 1 ? e : 1.0;
        const double a = num / den, b = div_unit_range_f64(num, den);
        if (__double_as_longlong(a) != __double_as_longlong(b)) {
            if (atomicAdd(mism, 1ull) == 0) { first[0] = (unsigned long long)__double_as_longlong(num); first[1] = (unsigned long long)__double_as_longlong(den); }
        }
    }
}

int main() {
    unsigned long long *mism, *first;
    hipMalloc(&mism, 8); hipMalloc(&first, 16);
    hipMemset(mism, 0, 8); hipMemset(first, 0, 16);
    const uint64_t n = 1ull << 32;
    check<<<256 * 32, 256>>>(mism, first, n);
    hipDeviceSynchronize();
    unsigned long long hm, hf[2];
    hipMemcpy(&hm, mism, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost);
    printf("div_unit_range_f64 vs IEEE '/' on 2^32 (num, 1 + e) pairs: %llu mismatches (first: num 0x%016llx den 0x%016llx)\n", hm, hf[0], hf[1]);
    return hm != 0;
}
