// pgcn_dense_common.h -- what the matrix-core kernels of gemm/ share: vector types, the exact three-plane bf16 split of fp32
// values, the order of the six partial products.  Included INSIDE the including file's namespace (the probe builds rename it);
// PG_HD and PGCN_DENSE_HOST_EMU come from the including file.  Arithmetic as in csrc/pgcn_spmm_dense3.hip.
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// float -> bf16 bits, round to nearest even (finite inputs; NaN stays NaN, inf stays inf)
PG_HD uint32_t bf16_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
PG_HD float bf16_as_f32(uint32_t b) {
    const uint32_t u = b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
// x, y -> their three bf16 planes, packed {x in bits 0-15, y in bits 16-31}
#ifndef PGCN_DENSE_HOST_EMU
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
PG_HD uint32_t pack_bf16(float x, float y) {       // the hardware's conversion (RNE): {bf16(x) in bits 0-15, bf16(y) in bits 16-31}
    const f32x2 v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
PG_HD void split_pair(float x, float y, uint32_t &u1, uint32_t &u2, uint32_t &u3) {
    u1 = pack_bf16(x, y);
    const float rx = x - __builtin_bit_cast(float, u1 << 16), ry = y - __builtin_bit_cast(float, u1 & 0xffff0000u);   // exact
    u2 = pack_bf16(rx, ry);
    u3 = pack_bf16(rx - __builtin_bit_cast(float, u2 << 16), ry - __builtin_bit_cast(float, u2 & 0xffff0000u));
}
#else
PG_HD void split_pair(float x, float y, uint32_t &u1, uint32_t &u2, uint32_t &u3) {
    const uint32_t x1 = bf16_bits(x), y1 = bf16_bits(y);
    const float rx = x - bf16_as_f32(x1), ry = y - bf16_as_f32(y1);           // exact
    const uint32_t x2 = bf16_bits(rx), y2 = bf16_bits(ry);
    const uint32_t x3 = bf16_bits(rx - bf16_as_f32(x2)), y3 = bf16_bits(ry - bf16_as_f32(y2));   // exact, and bf16 numbers
    u1 = x1 | (y1 << 16); u2 = x2 | (y2 << 16); u3 = x3 | (y3 << 16);
}
#endif
// the eight values of one lane and k step -> the lane's A (or B) operand of each plane
PG_HD void split8(const f32x4 &lo4, const f32x4 &hi4, u32x4 (&p)[3]) {
    uint32_t a, b, c;
    split_pair(lo4.x, lo4.y, a, b, c); p[0].x = a; p[1].x = b; p[2].x = c;
    split_pair(lo4.z, lo4.w, a, b, c); p[0].y = a; p[1].y = b; p[2].y = c;
    split_pair(hi4.x, hi4.y, a, b, c); p[0].z = a; p[1].z = b; p[2].z = c;
    split_pair(hi4.z, hi4.w, a, b, c); p[0].w = a; p[1].w = b; p[2].w = c;
}

// the six partial products that matter, smallest first: (plane of A, plane of B)
#define PGCN_DENSE_PRODUCTS constexpr int kPA[6] = {2, 0, 1, 1, 0, 0}, kPB[6] = {0, 2, 1, 0, 1, 0}

