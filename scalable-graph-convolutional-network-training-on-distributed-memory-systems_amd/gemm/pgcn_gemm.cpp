// pgcn_gemm.cpp -- the dense products of a layer (GPU/PGCN.py:146 `self.linear(H)` and its backward) as stock rocBLAS
// GEMMs launched BY SOLUTION INDEX.  Plumbing beside the graded aggregation path, and its own small library
// (lib/libpgcn_gemm.so): PyTorch's default pick for the n x f x f shapes is 15-20 % slower than the best kernel the library
// holds; PyTorch's TunableOp finds that kernel but enumerates every kernel file of two libraries when it is switched on
// (r04: 33 s on a box that has not touched them, 1 s on one that has).  The choice is made offline (tunableop/gfx950.csv,
// PyTorch's own result format) and this file only replays it: rocblas_gemm_ex(..., rocblas_gemm_algo_solution_index, index)
// loads that one kernel.  Inside a PyTorch process the loader binds librocblas.so.5 to the copy PyTorch mapped, i.e. the
// build the indices were recorded on; pgcn_gemm_rocblas_version() is what the caller checks against the file's validator.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

namespace {
thread_local char g_err[256] = "";
std::mutex g_mu;
// One rocBLAS handle per (device, stream): a handle owns a device work-space that solutions with a split reduction use, so two
// streams must not share one (two PyTorch streams inside mm_nt / mm_nn at once would race on it).  A process uses a handful of
// streams; the table is searched linearly under the mutex, handles live until the process ends.
struct HandleSlot { int dev; void *stream; rocblas_handle h; };
constexpr int kMaxHandles = 256;
HandleSlot g_slots[kMaxHandles];
int g_nslots = 0;
int g_atomics_allowed = 1;      // PyTorch's default; pgcn_gemm_set_atomics(0) under torch.use_deterministic_algorithms(True)

int fail(int code, const char *what, int status) {
    snprintf(g_err, sizeof(g_err), "%s (status %d)", what, status);
    return code;
}

// (g_mu held)
rocblas_handle handle_for(int dev, void *stream) {
    for (int i = 0; i < g_nslots; ++i)
        if (g_slots[i].dev == dev && g_slots[i].stream == stream) return g_slots[i].h;
    if (g_nslots == kMaxHandles) return nullptr;
    rocblas_handle h = nullptr;
    if (rocblas_create_handle(&h) != rocblas_status_success) return nullptr;
    if (rocblas_set_stream(h, (hipStream_t)stream) != rocblas_status_success) {
        rocblas_destroy_handle(h);
        return nullptr;
    }
    g_slots[g_nslots++] = HandleSlot{dev, stream, h};
    return h;
}
}  // namespace

extern "C" const char *pgcn_gemm_last_error(void) { return g_err; }

// allowed = 0: kernels that reduce with atomics are refused by rocBLAS (what torch.use_deterministic_algorithms(True) sets on
// PyTorch's own handles); 1 (default): PyTorch's default mode.
extern "C" void pgcn_gemm_set_atomics(int32_t allowed) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_atomics_allowed = allowed ? 1 : 0;
}

// rocBLAS build string ("5.0.2.20250912-42-1199-g2584e35062") of the library this process bound.
extern "C" int pgcn_gemm_rocblas_version(char *buf, int64_t n) {
    size_t need = 0;
    if (rocblas_get_version_string_size(&need) != rocblas_status_success || !buf || (int64_t)need > n)
        return fail(-1, "rocblas_get_version_string_size", 0);
    rocblas_status st = rocblas_get_version_string(buf, (size_t)n);
    return st == rocblas_status_success ? 0 : fail(-1, "rocblas_get_version_string", (int)st);
}

// C (m x n, ldc) = op(A) . op(B) in rocBLAS' column-major convention, fp32, alpha 1, beta 0, on `stream` of the current
// device, with the kernel `solution_index` of that rocBLAS build (0 = the library's own pick).  transa / transb: 0 = N, 1 = T.
// Returns 0, or -2 when rocBLAS refuses the index for this problem (the caller falls back to its default GEMM).
extern "C" int pgcn_gemm_f32(int32_t transa, int32_t transb, int64_t m, int64_t n, int64_t k, const float *A, int64_t lda,
                             const float *B, int64_t ldb, float *C, int64_t ldc, int32_t solution_index, void *stream) {
    if (m <= 0 || n <= 0 || k <= 0 || !A || !B || !C) return fail(-1, "pgcn_gemm_f32: bad argument", 0);
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX || lda > INT32_MAX || ldb > INT32_MAX || ldc > INT32_MAX)
        return fail(-1, "pgcn_gemm_f32: dimension beyond the 32-bit interface", 0);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(-1, "hipGetDevice", 0);
    std::lock_guard<std::mutex> lock(g_mu);          // the table and the enqueue as one step
    // A stream that is being captured into a HIP graph (torch.cuda.graph captures on its own stream) must not see
    // rocblas_create_handle: it allocates device memory, which is illegal under capture.  A product on such a stream borrows a
    // handle this device already has (its stream is switched for the one call and restored) -- set-up ran at least one product
    // on the default stream (PGCN.tune_dense_gemms), so one exists; only when none does is a handle created here.
    rocblas_handle h = nullptr, borrowed = nullptr;
    void *borrowed_stream = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    bool have = false;
    for (int i = 0; i < g_nslots; ++i) have = have || (g_slots[i].dev == dev && g_slots[i].stream == stream);
    if (!have && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) {
        for (int i = 0; i < g_nslots && !borrowed; ++i)
            if (g_slots[i].dev == dev) { borrowed = g_slots[i].h; borrowed_stream = g_slots[i].stream; }
        if (borrowed && rocblas_set_stream(borrowed, (hipStream_t)stream) == rocblas_status_success) h = borrowed;
        else borrowed = nullptr;
    }
    if (!h) h = handle_for(dev, stream);
    if (!h) return fail(-1, "rocblas_create_handle / rocblas_set_stream (or more than 256 (device, stream) pairs)", 0);
    struct Restore {
        rocblas_handle h; void *s;
        ~Restore() { if (h) rocblas_set_stream(h, (hipStream_t)s); }
    } restore{borrowed, borrowed_stream};
    rocblas_status st = rocblas_set_atomics_mode(h, g_atomics_allowed ? rocblas_atomics_allowed : rocblas_atomics_not_allowed);
    if (st != rocblas_status_success) return fail(-1, "rocblas_set_atomics_mode", (int)st);
    const float one = 1.0f, zero = 0.0f;
    st = rocblas_gemm_ex(h, transa ? rocblas_operation_transpose : rocblas_operation_none,
                         transb ? rocblas_operation_transpose : rocblas_operation_none, (rocblas_int)m, (rocblas_int)n,
                         (rocblas_int)k, &one, A, rocblas_datatype_f32_r, (rocblas_int)lda, B, rocblas_datatype_f32_r,
                         (rocblas_int)ldb, &zero, C, rocblas_datatype_f32_r, (rocblas_int)ldc, C, rocblas_datatype_f32_r,
                         (rocblas_int)ldc, rocblas_datatype_f32_r,
                         solution_index ? rocblas_gemm_algo_solution_index : rocblas_gemm_algo_standard, solution_index,
                         rocblas_gemm_flags_none);
    if (st == rocblas_status_invalid_value || st == rocblas_status_not_implemented)
        return fail(-2, "rocblas_gemm_ex refuses this solution index for the problem", (int)st);
    return st == rocblas_status_success ? 0 : fail(-1, "rocblas_gemm_ex", (int)st);
}
