// pgcn_dense_tile.h -- the index arithmetic of gemm/pgcn_dense.hip: where a value of W lands in the LDS image of B operands,
// which 16-byte pieces of a tile a lane loads, which element of C an accumulator register is, where a row's sign-mask words
// live, the argument checks.  Shared by
//   * gemm/pgcn_dense.hip (device build: PG_HD = __device__ __forceinline__), and
//   * tests/native/pgcn_dense_emu.cpp (host build, -DPGCN_DENSE_HOST_EMU: the same functions run lane by lane around an
//     emulated v_mfma_f32_32x32x16_bf16, so that slots, operand lanes, the accumulator layout and the mask bits are checked on the
//     CPU).
// Included INSIDE namespace pgcn_dense; PG_HD comes from the including file.
//
// r06 layout of the work (the kernel file has the schedule): a tile is handled PIECE by piece (k step ks: the lane's two 16-byte
// loads of row lo, columns 16 ks + 8 hi .. + 8) and its product column BLOCK by column block (nb: 16 accumulator registers =
// rows (r & 3) + 8 (r >> 2) + 4 hi, column 32 nb + lo), so that loads, splits and stores can be spread between the MFMAs.
// Rows beyond n are never predicated: every streamed matrix is addressed through a WINDOW that starts at the tile's first row and
// ends with the matrix (a buffer descriptor on the device): loads beyond it read zero, stores beyond it are dropped.  Widths that do not fill the instance (K < 16 NKS, N < 32 NBLK) take the RAGGED
// instantiations: pieces beyond K read as zero, columns beyond N are not stored.
constexpr int kRows = 32;                 // rows of a wave's tile = M of the MFMA
constexpr int kMaxF = 128;                // K and N of a product
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kSlotsPerPlane = 8 * 4 * 64;                 // (k step, column block, lane)
constexpr int kPlaneBytes = kSlotsPerPlane * 16;           // 32 KB
constexpr int kImageBytes = 3 * kPlaneBytes;               // 96 KB

#include "pgcn_dense_common.h"

// byte offset of a lane's B operand in the image
PG_HD int image_offset(int plane, int ks, int nb, int lane) { return plane * kPlaneBytes + ((ks * 4 + nb) * 64 + lane) * 16; }

// Slot s (0 .. kSlotsPerPlane) of the image, all three planes: the eight values Bm[16 ks + 8 hi + j][32 nb + lo].
// transposed = 1: Bm = W^T (W is N x K, row-major, ldw); 0: Bm = W (W is K x N).  Outside K x N: zeros.
// In two steps, so that a thread's loads of all its slots are in flight together (branch-free: clamped addresses, the value
// dropped afterwards).  `vec` (uniform: transposed, K % 8 == 0, ldw % 4 == 0, W 16-byte aligned): the eight values are eight
// consecutive floats of one row of W -- two 16-byte loads instead of eight dwords that each touch 32 cache lines per wave (r06: the
// image build of a workgroup drops from ~8 k to ~2 k line look-ups).
PG_HD void slot_load(const float *__restrict__ W, int64_t ldw, int transposed, int K, int N, int s, float (&v)[8], bool vec = false) {
    const int lane = s & 63, nb = (s >> 6) & 3, ks = s >> 8;
    const int col = 32 * nb + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    const int colc = col < N ? col : N - 1;
    if (vec) {                                            // (uniform) transposed: W[col][k0 .. k0 + 8)
        const bool in = k0 + 8 <= K;
        const float *p = W + (int64_t)colc * ldw + (in ? k0 : 0);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + 4);
        const bool ok = in && col < N;
        v[0] = ok ? a.x : 0.f; v[1] = ok ? a.y : 0.f; v[2] = ok ? a.z : 0.f; v[3] = ok ? a.w : 0.f;
        v[4] = ok ? b.x : 0.f; v[5] = ok ? b.y : 0.f; v[6] = ok ? b.z : 0.f; v[7] = ok ? b.w : 0.f;
        return;
    }
    const int64_t sk = transposed ? 1 : ldw, sc = transposed ? ldw : 1;      // (one address, one load: no branch on the mode)
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j, kc = k < K ? k : K - 1;
        x[j] = W[kc * sk + colc * sc];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (k0 + j < K && col < N) ? x[j] : 0.f;
}
// the vector form of slot_load applies (uniform per launch)
inline bool slot_vec_ok(const void *W, int64_t ldw, int transposed, int K) {
    return transposed && K % 8 == 0 && ldw % 4 == 0 && (uintptr_t)W % 16 == 0;
}
PG_HD void slot_store(char *image, int s, const float (&v)[8]) {
    const int lane = s & 63, nb = (s >> 6) & 3, ks = s >> 8;
    u32x4 p[3];
    const f32x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
    split8(lo4, hi4, p);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(image + image_offset(pl, ks, nb, lane)) = p[pl];
}

// ---- windows: the rows of a matrix from a tile's first row on, addressed as (window, lane offset, scalar offset) -----------------------
// On the device a window is a buffer descriptor (base = the tile's first row, size = the bytes from there to the end of the matrix):
// a lane's byte offset inside the tile never changes, the scalar offset picks the row of an accumulator register, and an access
// beyond the last row of the matrix is dropped (stores) or reads as zero (loads) by the hardware's bounds check -- no row predicates,
// no 64-bit address arithmetic in the loop.  The host build checks the same bounds in software.
#ifndef PGCN_DENSE_HOST_EMU
using window_t = __amdgpu_buffer_rsrc_t;
// (every argument wave-uniform; the caller passes values it made uniform with readfirstlane)
PG_HD window_t make_window(const void *base, int64_t bytes) {
    const uint32_t nb = bytes <= 0 ? 0u : (bytes > 0xfffff000LL ? 0xfffff000u : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)nb, 0x00020000);
}
PG_HD f32x4 win_load16(window_t w, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w, (int)voff, (int)soff, 0));
}
PG_HD uint32_t win_load4u(window_t w, uint32_t voff, uint32_t soff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(w, (int)voff, (int)soff, 0);
}
PG_HD void win_store16(window_t w, uint32_t voff, uint32_t soff, const f32x4 &v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), w, (int)voff, (int)soff, 0);
}
PG_HD void win_store4(window_t w, uint32_t voff, uint32_t soff, float v) {           // non-temporal: C is not read again by this kernel
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), w, (int)voff, (int)soff, 2);
}
PG_HD void win_store4u(window_t w, uint32_t voff, uint32_t soff, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, w, (int)voff, (int)soff, 0);
}
#else
struct window_t {
    char *base;
    int64_t bytes;
};
PG_HD window_t make_window(const void *base, int64_t bytes) {
    return window_t{const_cast<char *>(static_cast<const char *>(base)), bytes <= 0 ? 0 : (bytes > 0xfffff000LL ? 0xfffff000LL : bytes)};
}
PG_HD f32x4 win_load16(window_t w, uint32_t voff, uint32_t soff) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int64_t o = (int64_t)voff + soff;
    if (o + 16 <= w.bytes) memcpy(&v, w.base + o, 16);
    return v;
}
PG_HD uint32_t win_load4u(window_t w, uint32_t voff, uint32_t soff) {
    uint32_t v = 0;
    const int64_t o = (int64_t)voff + soff;
    if (o + 4 <= w.bytes) memcpy(&v, w.base + o, 4);
    return v;
}
PG_HD void win_store16(window_t w, uint32_t voff, uint32_t soff, const f32x4 &v) {
    const int64_t o = (int64_t)voff + soff;
    if (o + 16 <= w.bytes) memcpy(w.base + o, &v, 16);
}
PG_HD void win_store4(window_t w, uint32_t voff, uint32_t soff, float v) {
    const int64_t o = (int64_t)voff + soff;
    if (o + 4 <= w.bytes) memcpy(w.base + o, &v, 4);
}
PG_HD void win_store4u(window_t w, uint32_t voff, uint32_t soff, uint32_t v) {
    const int64_t o = (int64_t)voff + soff;
    if (o + 4 <= w.bytes) memcpy(w.base + o, &v, 4);
}
#endif
// the window of a row-major matrix M (rows `ld` elements apart, n rows) from row `row0` on; elem = bytes per element
PG_HD window_t tile_window(const void *M, int64_t ld, int64_t n, int64_t row0, int width, int elem = 4) {
    // the last row ends at its width, not at the next row's start (the matrix may sit at the very end of an allocation)
    const int64_t rows = n - row0;
    return make_window(static_cast<const char *>(M) + row0 * ld * elem, rows <= 0 ? 0 : ((rows - 1) * ld + width) * elem);
}

// ---- a lane's pieces of a tile ----------------------------------------------------------------------------------------------
// byte offset of the lane's row inside a tile's window, at the lane's first column of a k step: row lo, column 8 hi
PG_HD uint32_t piece_lane_offset(int64_t ld, int lane) { return (uint32_t)(((int64_t)(lane & 31) * ld + 8 * (lane >> 5)) * 4); }
// piece ks of the lane's row: A[row][16 ks + 8 hi + 4 h .. + 4], h = 0, 1.  RAGGED: pieces at or beyond K are zero (K % 4 == 0: a piece
// lies inside or outside the width as a whole).  Rows beyond the matrix read as zero (the window's bounds).
template <bool RAGGED>
PG_HD void load_piece(f32x4 (&v)[2], window_t w, uint32_t lane_off, int ks, int K, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if constexpr (!RAGGED) {
            v[h] = win_load16(w, lane_off + (uint32_t)(64 * ks + 16 * h), 0);
        } else {
            const bool ok = 16 * ks + 8 * (lane >> 5) + 4 * h < K;
            const f32x4 x = win_load16(w, ok ? lane_off + (uint32_t)(64 * ks + 16 * h) : lane_off - (uint32_t)(32 * (lane >> 5)), 0);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v[h] = ok ? x : z;
        }
    }
}
// the same piece written back (Gm of the input gradient): rows beyond the matrix are dropped by the window, RAGGED: pieces beyond K by
// the lane
template <bool RAGGED>
PG_HD void store_piece(const f32x4 (&v)[2], window_t w, uint32_t lane_off, int ks, int K, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (!RAGGED || 16 * ks + 8 * (lane >> 5) + 4 * h < K) win_store16(w, lane_off + (uint32_t)(64 * ks + 16 * h), 0, v[h]);
    }
}

// ---- the sign mask of the forward's output (r06) -----------------------------------------------------------------------------------
// mask[row * mw + w] bit b = (Y[row][32 w + b] > 0), mw = words per row (mask_words(N)): what the input gradient needs of Y -- 4 bytes
// per row and 32 columns instead of 128 (the backward reads 3.7 MB instead of 119 MB at the benchmark layer, and holds 4 registers of it
// per tile instead of 64).
PG_HD int mask_words(int N) { return (N + 31) / 32; }
// The lane that assembles the mask word of a tile row for a column block: accumulator register r of half wave h holds row
// (r & 3) + 8 (r >> 2) + 4 h; the ballot of (acc[r] > 0) has that row's 32 column bits of half h in bits 32 h .. 32 h + 31.
// The word of (r, h) is collected in lane 16 h + r: mask_row_of_lane(L) is the tile row whose word lane L (< 32) holds.
PG_HD int mask_row_of_lane(int L) {
    const int r = L & 15, h = L >> 4;
    return (r & 3) + 8 * (r >> 2) + 4 * h;
}
// threshold_backward by the mask bits: piece (ks, h) of a row covers columns 16 ks + 8 hi + 4 h .. + 4 = bits (16 (ks & 1) + 8 hi + 4 h) ..
// of word ks >> 1
PG_HD f32x4 mask4_bits(const f32x4 &g, uint32_t word, int ks, int h, int lane) {
    const uint32_t sh = word >> (16 * (ks & 1) + 8 * (lane >> 5) + 4 * h);
    f32x4 r;
    r.x = (sh & 1u) ? g.x : 0.f; r.y = (sh & 2u) ? g.y : 0.f; r.z = (sh & 4u) ? g.z : 0.f; r.w = (sh & 8u) ? g.w : 0.f;
    return r;
}
// threshold_backward(g, y, 0): the gradient where y > 0 (NaN in y keeps it, like ATen's `y <= 0 ? 0 : g`)
PG_HD f32x4 mask4(const f32x4 &g, const f32x4 &y) {
    f32x4 r;
    r.x = y.x <= 0.f ? 0.f : g.x; r.y = y.y <= 0.f ? 0.f : g.y; r.z = y.z <= 0.f ? 0.f : g.z; r.w = y.w <= 0.f ? 0.f : g.w;
    return r;
}
// (the mask's convention for NaN: `!(y <= 0)`, the same predicate as mask4)
PG_HD bool mask_bit_of(float y) { return !(y <= 0.f); }
PG_HD float relu1(float x) { return x < 0.f ? 0.f : x; }                      // clamp_min(0): NaN stays NaN

// byte offset of a lane's first accumulator row inside a tile's window of C: row 4 hi, column lo
PG_HD uint32_t acc_lane_offset(int64_t ldc, int lane) { return (uint32_t)(((int64_t)(4 * (lane >> 5)) * ldc + (lane & 31)) * 4); }
// registers [r0, r0 + cnt) of column block nb: accumulator register r of lane (lo, hi) -> element (row0 + (r & 3) + 8 (r >> 2) + 4 hi,
// 32 nb + lo) of C = window of the tile, lane offset + 128 nb, scalar offset ((r & 3) + 8 (r >> 2)) rows.  Rows beyond the matrix are
// dropped by the window; RAGGED: columns beyond N by the lane.
template <bool RAGGED>
PG_HD void store_regs(const f32x16 &acc, int r0, int cnt, int nb, window_t wc, uint32_t lane_off, uint32_t row_bytes, int N, int lane,
                      int relu) {
    if (RAGGED && 32 * nb + (lane & 31) >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (r >= r0 && r < r0 + cnt)
            win_store4(wc, lane_off + (uint32_t)(128 * nb), (uint32_t)((r & 3) + 8 * (r >> 2)) * row_bytes, relu ? relu1(acc[r]) : acc[r]);
}

// ---- argument checks and the error string ---------------------------------------------------------------------------------
inline thread_local char g_err[256] = "";
inline int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

inline int check(const void *A, int64_t lda, int64_t n, int K, int N, const void *W, int64_t ldw, int wrows, int wcols,
                 const void *C, int64_t ldc) {
    if (n < 0 || K <= 0 || N <= 0 || !W || (n > 0 && (!A || !C))) return fail(-1, "pgcn_dense: bad argument");
    if (K > kMaxF || N > kMaxF) return fail(-2, "pgcn_dense: widths above 128 are left to the library GEMM");
    if (K % 4 || lda % 4 || (uintptr_t)A % 16) return fail(-2, "pgcn_dense: rows of the left operand must be 16-byte pieces");
    if (lda < K || ldc < N || ldw < wcols || wrows <= 0) return fail(-1, "pgcn_dense: leading dimension below the width");
    if (n > ((int64_t)1 << 40)) return fail(-1, "pgcn_dense: n out of range");
    return 0;
}
