/* Exhaustive check of the three-instruction division csrc/roi_align.hip uses for "accumulator / sample count":
 *     y = RN(1 / c);  q = RN(x * y);  result = RN(q + RN(fma(-c, q, x)) * y)        (the residual is exact in an FMA)
 * against IEEE division, for every count c = g1 * g2 with g1, g2 <= 22 (the kernel divides beyond that) and every one of the
 * 2^23 float significands (scaling by powers of two changes neither side as long as nothing underflows; accumulators below
 * 2^-25 round to fp16 zero either way).  Prints the number of (count, significand) pairs that differ: must be 0. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
    static char seen[512];
    long bad = 0, counts = 0;
    for (int g1 = 1; g1 <= 22; ++g1)
        for (int g2 = g1; g2 <= 22; ++g2) {
            const int ci = g1 * g2;
            if (seen[ci]) continue;
            seen[ci] = 1;
            ++counts;
            volatile float c = (float)ci;
            const float y = 1.0f / c;
            for (uint32_t m = 0; m < (1u << 23); ++m) {
                const uint32_t bits = 0x3f800000u | m;
                float x;
                memcpy(&x, &bits, 4);
                const float q = x * y;
                const float r = fmaf(-c, q, x);
                const float got = fmaf(r, y, q);
                bad += got != x / c;
            }
        }
    printf("%ld %ld\n", counts, bad);
    return 0;
}
