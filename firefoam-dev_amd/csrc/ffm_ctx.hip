// ffm_ctx.hip -- context, error reporting, device-memory helpers, reductions.
#include "ffm_internal.hpp"
#include <algorithm>
#include "ffm_device.hpp"
#include <cstdarg>

static thread_local char g_err[1024] = "";

void ffm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ffm_last_error(void) { return g_err; }
extern "C" const char *ffm_version(void) { return "ffm 0.1 (gfx950, fp64/int32, fp-contract=off)"; }

extern "C" int ffm_ctx_create(int device, void *stream, ffm_ctx **out)
{
    if (!out) return FFM_ERR_ARG;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) {
        ffm_set_error("ffm_ctx_create: no HIP device visible (the HIP path has no CPU fallback)");
        return FFM_ERR_NODEVICE;
    }
    if (device < 0 || device >= nDev) { ffm_set_error("ffm_ctx_create: device %d of %d", device, nDev); return FFM_ERR_ARG; }
    FFM_HIP(hipSetDevice(device));
    ffm_ctx *c = new ffm_ctx();
    c->device = device;
    if (stream) { c->stream = (hipStream_t)stream; c->ownStream = false; }
    else { FFM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->ownStream = true; }
    FFM_HIP(hipMalloc((void **)&c->scal_d, sizeof(double) * NSCAL));
    FFM_HIP(hipMemsetAsync(c->scal_d, 0, sizeof(double) * NSCAL, c->stream));
    FFM_HIP(hipMalloc((void **)&c->partials_d, sizeof(double) * 4 * RED_BLOCKS));
    FFM_HIP(hipHostMalloc((void **)&c->scal_h, sizeof(double) * NSCAL, hipHostMallocDefault));
    hipDeviceProp_t prop;
    FFM_HIP(hipGetDeviceProperties(&prop, device));
    c->cuCount = prop.multiProcessorCount;
    *out = c;
    return FFM_OK;
}

extern "C" int ffm_ctx_destroy(ffm_ctx *c)
{
    if (!c) return FFM_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    ffm_comm_finalize_i(c);
    ffm_ctx_trim(c);
    hipFree(c->scal_d); hipFree(c->partials_d); hipHostFree(c->scal_h);
    if (c->multiScal_d) { hipFree(c->multiScal_d); hipHostFree(c->multiScal_h); }
    if (c->ownStream) hipStreamDestroy(c->stream);
    delete c;
    return FFM_OK;
}

extern "C" int ffm_ctx_sync(ffm_ctx *c) { FFM_HIP(hipStreamSynchronize(c->stream)); return FFM_OK; }
extern "C" void *ffm_ctx_stream(ffm_ctx *c) { return (void *)c->stream; }

static inline size_t pool_class(size_t bytes) { return bytes <= 256 ? 256 : ((bytes + 4095) & ~(size_t)4095); }
extern "C" int ffm_ctx_trim(ffm_ctx *c)
{
    if (!c) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(c->device));
    FFM_HIP(hipStreamSynchronize(c->stream));
    for (auto &kv : c->poolFree) for (void *q : kv.second) { c->poolSize.erase(q); hipFree(q); }
    c->poolFree.clear(); c->poolCachedBytes = 0;
    return FFM_OK;
}
__global__ void k_zero_bytes8(size_t n8, unsigned long long *__restrict__ d)
{ for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) d[i] = 0ull; }
// zero-fill on the context's stream by a kernel of this library (size classes are multiples of 8 bytes)
static int pool_zero(ffm_ctx *c, void *p, size_t bytes)
{
    const size_t n8 = bytes / 8;
    if (n8) hipLaunchKernelGGL(k_zero_bytes8, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n8 + 255) / 256, 4096))), dim3(256), 0, c->stream, n8, (unsigned long long *)p);
    if (bytes % 8) FFM_HIP(hipMemsetAsync((char *)p + 8 * n8, 0, bytes % 8, c->stream));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
static int pool_alloc(ffm_ctx *c, size_t bytes, void **p, bool zero)
{
    if (!c || !p) return FFM_ERR_ARG;
    const size_t cls = pool_class(bytes);
    auto it = c->poolFree.find(cls);
    if (it != c->poolFree.end() && !it->second.empty()) {
        *p = it->second.back(); it->second.pop_back(); c->poolCachedBytes -= cls;
        if (zero) FFM_TRY(pool_zero(c, *p, cls));          // as fresh memory: the padding slots of face arrays must read 0
        return FFM_OK;
    }
    FFM_HIP(hipSetDevice(c->device));
    if (hipMalloc(p, cls) != hipSuccess) {                       // out of memory: give the cached blocks back and try once more
        (void)hipGetLastError();
        FFM_TRY(ffm_ctx_trim(c));
        FFM_HIP(hipMalloc(p, cls));
    }
    c->poolSize[*p] = cls;
    if (zero) FFM_TRY(pool_zero(c, *p, cls));
    return FFM_OK;
}
extern "C" int ffm_malloc(ffm_ctx *c, size_t bytes, void **p) { return pool_alloc(c, bytes, p, true); }
extern "C" int ffm_malloc_uninit(ffm_ctx *c, size_t bytes, void **p) { return pool_alloc(c, bytes, p, false); }
extern "C" int ffm_free(ffm_ctx *c, void *p)
{
    if (!c) return FFM_ERR_ARG;
    if (!p) return FFM_OK;
    auto it = c->poolSize.find(p);
    if (it == c->poolSize.end()) { FFM_HIP(hipStreamSynchronize(c->stream)); FFM_HIP(hipFree(p)); return FFM_OK; }   // not ours
    if (c->poolCachedBytes + it->second > c->poolCapBytes) {
        FFM_HIP(hipStreamSynchronize(c->stream)); FFM_HIP(hipFree(p)); c->poolSize.erase(it); return FFM_OK;
    }
    c->poolFree[it->second].push_back(p); c->poolCachedBytes += it->second;
    return FFM_OK;
}
extern "C" int ffm_memcpy_h2d(ffm_ctx *c, void *d, const void *s, size_t n)
{
    // Every host <-> device copy of the layer goes through the context's stream and is waited for: blocks handed out by
    // ffm_malloc carry a zero-fill queued on that (non-blocking) stream, which a null-stream copy does not wait for.
    FFM_HIP(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, c->stream)); FFM_HIP(hipStreamSynchronize(c->stream)); return FFM_OK;
}
extern "C" int ffm_memcpy_d2h(ffm_ctx *c, void *d, const void *s, size_t n)
{ FFM_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, c->stream)); FFM_HIP(hipStreamSynchronize(c->stream)); return FFM_OK; }
extern "C" int ffm_memcpy_d2d(ffm_ctx *c, void *d, const void *s, size_t n)
{ FFM_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, c->stream)); return FFM_OK; }
extern "C" int ffm_memset(ffm_ctx *c, void *d, int v, size_t n)
{ FFM_HIP(hipMemsetAsync(d, v, n, c->stream)); return FFM_OK; }

int ffm_read_scalars(ffm_ctx *c)
{
    FFM_HIP(hipMemcpyAsync(c->scal_h, c->scal_d, sizeof(double) * NSCAL, hipMemcpyDeviceToHost, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    return FFM_OK;
}

// ------------------------------------------------------------- reductions ---
// Stage 1: RED_BLOCKS blocks, grid-stride, per-block partial in fixed order.
// Stage 2: one block sums the partials in fixed order.  No atomics anywhere.
enum { R_SUM = 0, R_DOT, R_SUMMAG, R_SUMSQR, R_MIN, R_MAX };

template <int OP>
__global__ __launch_bounds__(RED_THREADS) void k_reduce1(long n, const double *__restrict__ x,
                                                         const double *__restrict__ y,
                                                         double *__restrict__ partials)
{
    __shared__ double sm[RED_THREADS / 64];
    double acc = (OP == R_MIN) ? 1.79769313486231570e+308 : (OP == R_MAX) ? -1.79769313486231570e+308 : 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double v = x[i];
        if (OP == R_SUM) acc += v;
        else if (OP == R_DOT) acc += v * y[i];
        else if (OP == R_SUMMAG) acc += fabs(v);
        else if (OP == R_SUMSQR) acc += v * v;
        else if (OP == R_MIN) acc = fmin(acc, v);
        else acc = fmax(acc, v);
    }
    double r;
    if (OP == R_MIN) r = block_min(acc, sm);
    else if (OP == R_MAX) r = block_max(acc, sm);
    else r = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

template <int OP>
__global__ __launch_bounds__(1024) void k_reduce2(int nPartials, const double *__restrict__ partials,
                                                  double *__restrict__ scal, int slot)
{
    __shared__ double sm[16];
    double acc = (OP == R_MIN) ? 1.79769313486231570e+308 : (OP == R_MAX) ? -1.79769313486231570e+308 : 0.0;
    for (int i = threadIdx.x; i < nPartials; i += blockDim.x) {
        double v = partials[i];
        if (OP == R_MIN) acc = fmin(acc, v);
        else if (OP == R_MAX) acc = fmax(acc, v);
        else acc += v;
    }
    double r;
    if (OP == R_MIN) r = block_min(acc, sm);
    else if (OP == R_MAX) r = block_max(acc, sm);
    else r = block_sum(acc, sm);
    if (threadIdx.x == 0) scal[slot] = r;
}

template <int OP>
static int reduce_to_slot(ffm_ctx *c, const double *x, const double *y, long n, int slot)
{
    int nb = (int)((n + RED_THREADS - 1) / RED_THREADS);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_reduce1<OP>, dim3(nb), dim3(RED_THREADS), 0, c->stream, n, x, y, c->partials_d);
    constexpr int OP2 = (OP == R_MIN) ? R_MIN : (OP == R_MAX) ? R_MAX : R_SUM;
    hipLaunchKernelGGL(k_reduce2<OP2>, dim3(1), dim3(1024), 0, c->stream, nb, c->partials_d, c->scal_d, slot);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

int ffm_k_dot(ffm_ctx *c, const double *x, const double *y, long n, int slot) { return reduce_to_slot<R_DOT>(c, x, y, n, slot); }
int ffm_k_summag(ffm_ctx *c, const double *x, long n, int slot) { return reduce_to_slot<R_SUMMAG>(c, x, nullptr, n, slot); }
int ffm_k_sum(ffm_ctx *c, const double *x, long n, int slot) { return reduce_to_slot<R_SUM>(c, x, nullptr, n, slot); }
int ffm_k_sumsqr(ffm_ctx *c, const double *x, long n, int slot) { return reduce_to_slot<R_SUMSQR>(c, x, nullptr, n, slot); }

template <int OP>
static int reduce_public(ffm_ctx *c, const double *x, const double *y, long n, double *out)
{
    if (!c || !out || n < 0) return FFM_ERR_ARG;
    FFM_TRY(reduce_to_slot<OP>(c, x, y, n, S_TMP0));
    if (c->nRanks > 1 || c->comm) {
        if (OP == R_MIN) FFM_TRY(ffm_allreduce_minmax(c, S_TMP0, 0));
        else if (OP == R_MAX) FFM_TRY(ffm_allreduce_minmax(c, S_TMP0, 1));
        else FFM_TRY(ffm_allreduce_slots(c, S_TMP0, 1));
    }
    FFM_TRY(ffm_read_scalars(c));
    *out = c->scal_h[S_TMP0];
    return FFM_OK;
}

extern "C" int ffm_reduce_sum(ffm_ctx *c, const double *x, long n, double *o) { return reduce_public<R_SUM>(c, x, nullptr, n, o); }
extern "C" int ffm_reduce_min(ffm_ctx *c, const double *x, long n, double *o) { return reduce_public<R_MIN>(c, x, nullptr, n, o); }
extern "C" int ffm_reduce_max(ffm_ctx *c, const double *x, long n, double *o) { return reduce_public<R_MAX>(c, x, nullptr, n, o); }
extern "C" int ffm_reduce_dot(ffm_ctx *c, const double *x, const double *y, long n, double *o) { return reduce_public<R_DOT>(c, x, y, n, o); }
extern "C" int ffm_reduce_summag(ffm_ctx *c, const double *x, long n, double *o) { return reduce_public<R_SUMMAG>(c, x, nullptr, n, o); }

// ------------------------------------------------------------------ element-wise field algebra ---
template <class F> __global__ void k_field(long n, F f, double *__restrict__ out)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = f(i);
}
template <class F> static int field_launch(ffm_ctx *c, long n, F f, double *out)
{
    if (!c || n < 0 || (n && !out)) return FFM_ERR_ARG;
    if (n == 0) return FFM_OK;
    FFM_HIP(hipSetDevice(c->device));
    const int g = (int)std::max(1L, std::min((n + 255) / 256, (long)RED_BLOCKS));
    hipLaunchKernelGGL(k_field<F>, dim3(g), dim3(256), 0, c->stream, n, f, out);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_field_binary(ffm_ctx *c, int op, long n, const double *a, const double *b, double *out)
{
    if (n && (!a || !b)) return FFM_ERR_ARG;
    switch (op) {
    case FFM_OP_ADD: return field_launch(c, n, [=] __device__(long i) { return a[i] + b[i]; }, out);
    case FFM_OP_SUB: return field_launch(c, n, [=] __device__(long i) { return a[i] - b[i]; }, out);
    case FFM_OP_MUL: return field_launch(c, n, [=] __device__(long i) { return a[i] * b[i]; }, out);
    case FFM_OP_DIV: return field_launch(c, n, [=] __device__(long i) { return a[i] / b[i]; }, out);
    case FFM_OP_MAX: return field_launch(c, n, [=] __device__(long i) { return fmax(a[i], b[i]); }, out);
    case FFM_OP_MIN: return field_launch(c, n, [=] __device__(long i) { return fmin(a[i], b[i]); }, out);
    case FFM_OP_NEGSEL: return field_launch(c, n, [=] __device__(long i) { return a[i] < 0.0 ? b[i] : a[i]; }, out);
    }
    return FFM_ERR_ARG;
}
extern "C" int ffm_field_scalar(ffm_ctx *c, int op, long n, const double *a, double s, int sf, double *out)
{
    if (n && !a) return FFM_ERR_ARG;
    switch (op) {
    case FFM_OP_ADD: return field_launch(c, n, [=] __device__(long i) { return sf ? s + a[i] : a[i] + s; }, out);
    case FFM_OP_SUB: return field_launch(c, n, [=] __device__(long i) { return sf ? s - a[i] : a[i] - s; }, out);
    case FFM_OP_MUL: return field_launch(c, n, [=] __device__(long i) { return sf ? s * a[i] : a[i] * s; }, out);
    case FFM_OP_DIV: return field_launch(c, n, [=] __device__(long i) { return sf ? s / a[i] : a[i] / s; }, out);
    case FFM_OP_MAX: return field_launch(c, n, [=] __device__(long i) { return fmax(a[i], s); }, out);
    case FFM_OP_MIN: return field_launch(c, n, [=] __device__(long i) { return fmin(a[i], s); }, out);
    }
    return FFM_ERR_ARG;
}
extern "C" int ffm_field_unary(ffm_ctx *c, int op, long n, const double *a, double *out)
{
    if (n && !a) return FFM_ERR_ARG;
    switch (op) {
    case FFM_UN_NEG: return field_launch(c, n, [=] __device__(long i) { return -a[i]; }, out);
    case FFM_UN_SQR: return field_launch(c, n, [=] __device__(long i) { return a[i] * a[i]; }, out);
    case FFM_UN_MAG: return field_launch(c, n, [=] __device__(long i) { return fabs(a[i]); }, out);
    case FFM_UN_SQRT: return field_launch(c, n, [=] __device__(long i) { return sqrt(a[i]); }, out);
    case FFM_UN_POS0: return field_launch(c, n, [=] __device__(long i) { return a[i] >= 0.0 ? 1.0 : 0.0; }, out);
    }
    return FFM_ERR_ARG;
}
extern "C" int ffm_field_fill(ffm_ctx *c, long n, double s, double *out)
{ return field_launch(c, n, [=] __device__(long) { return s; }, out); }

// One element-wise expression in one pass: a postfix program over a four-deep operand stack.  Every operation is the one the
// separate ffm_field_binary / _scalar / _unary call performs (same operand order, one rounding each; the library is compiled with
// -ffp-contract=off), so the result is bit for bit that of the chain of calls -- without the intermediate arrays.  The operands of an
// element are loaded up front (all loads in flight together), the program then runs on registers; its branches are wave-uniform.
struct EvalProgram { int nInstr; const double *arr[FFM_EVAL_MAX_ARRAYS]; double imm[FFM_EVAL_MAX_IMM]; int code[FFM_EVAL_MAX_INSTR]; };
__device__ __forceinline__ double eval_binary(int op, double a, double b)
{
    switch (op) {
    case FFM_OP_ADD: return a + b;
    case FFM_OP_SUB: return a - b;
    case FFM_OP_MUL: return a * b;
    case FFM_OP_DIV: return a / b;
    case FFM_OP_MAX: return fmax(a, b);
    case FFM_OP_MIN: return fmin(a, b);
    default: return a < 0.0 ? b : a;          // FFM_OP_NEGSEL
    }
}
__device__ __forceinline__ double eval_unary(int op, double a)
{
    switch (op) {
    case FFM_UN_NEG: return -a;
    case FFM_UN_SQR: return a * a;
    case FFM_UN_MAG: return fabs(a);
    case FFM_UN_SQRT: return sqrt(a);
    default: return a >= 0.0 ? 1.0 : 0.0;     // FFM_UN_POS0
    }
}
// NA = number of operand arrays (1..8): the loads and the operand selection are unrolled over it, nothing is indexed dynamically
template <int NA>
__global__ void __launch_bounds__(256) k_field_eval(long n, EvalProgram P, double *__restrict__ out)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 2 * stride) {
        const long j = i + stride < n ? i + stride : i;          // the second element of this pass (the first again at the very end)
#define EV_LD(k) const double x##k = k < NA ? P.arr[k < NA ? k : 0][i] : 0.0, y##k = k < NA ? P.arr[k < NA ? k : 0][j] : 0.0;
        EV_LD(0) EV_LD(1) EV_LD(2) EV_LD(3) EV_LD(4) EV_LD(5) EV_LD(6) EV_LD(7)
#undef EV_LD
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        for (int k = 0; k < P.nInstr; k++) {
            const int c = P.code[k], kind = c >> 12, arg = c & 0xfff;
            if (kind == FFM_EVAL_LOAD || kind == FFM_EVAL_IMM) {
                s3 = s2; s2 = s1; s1 = s0; t3 = t2; t2 = t1; t1 = t0;
                if (kind == FFM_EVAL_IMM) s0 = t0 = P.imm[arg];
                else {
                    s0 = x0; t0 = y0;
#define EV_PICK(k) if (k < NA && arg == k) { s0 = x##k; t0 = y##k; }
                    EV_PICK(1) EV_PICK(2) EV_PICK(3) EV_PICK(4) EV_PICK(5) EV_PICK(6) EV_PICK(7)
#undef EV_PICK
                }
            } else if (kind == FFM_EVAL_BINARY) {
                s0 = eval_binary(arg, s1, s0); t0 = eval_binary(arg, t1, t0);
                s1 = s2; s2 = s3; t1 = t2; t2 = t3;
            } else { s0 = eval_unary(arg, s0); t0 = eval_unary(arg, t0); }
        }
        out[i] = s0;
        if (j != i) out[j] = t0;
    }
}
extern "C" int ffm_field_eval(ffm_ctx *c, long n, int nArrays, const double *const *arrays, int nImm, const double *imm, int nInstr,
                              const unsigned short *code, double *out)
{
    if (!c || n < 0 || (n && !out) || nArrays < 0 || nArrays > FFM_EVAL_MAX_ARRAYS || nImm < 0 || nImm > FFM_EVAL_MAX_IMM || nInstr < 1 ||
        nInstr > FFM_EVAL_MAX_INSTR || !code || (nArrays && !arrays) || (nImm && !imm))
        return FFM_ERR_ARG;
    // the program must leave exactly one value and never hold more than four
    int depth = 0;
    for (int k = 0; k < nInstr; k++) {
        const int kind = code[k] >> 12, arg = code[k] & 0xfff;
        if (kind == FFM_EVAL_LOAD) { if (arg >= nArrays || (n && !arrays[arg])) return FFM_ERR_ARG; depth++; }
        else if (kind == FFM_EVAL_IMM) { if (arg >= nImm) return FFM_ERR_ARG; depth++; }
        else if (kind == FFM_EVAL_BINARY) { if (arg > FFM_OP_NEGSEL || depth < 2) return FFM_ERR_ARG; depth--; }
        else if (kind == FFM_EVAL_UNARY) { if (arg > FFM_UN_POS0 || depth < 1) return FFM_ERR_ARG; }
        else return FFM_ERR_ARG;
        if (depth > 4) return FFM_ERR_ARG;
    }
    if (depth != 1) return FFM_ERR_ARG;
    if (n == 0) return FFM_OK;
    if (nArrays == 0) return FFM_ERR_ARG;          // a constant is ffm_field_fill
    EvalProgram P{};
    P.nInstr = nInstr;
    for (int k = 0; k < nArrays; k++) P.arr[k] = arrays[k];
    for (int k = 0; k < nImm; k++) P.imm[k] = imm[k];
    for (int k = 0; k < nInstr; k++) P.code[k] = code[k];
    FFM_HIP(hipSetDevice(c->device));
    const int g = (int)std::max(1L, std::min((n + 511) / 512, (long)RED_BLOCKS));
    switch (nArrays) {
#define EV_CASE(NA) case NA: hipLaunchKernelGGL(k_field_eval<NA>, dim3(g), dim3(256), 0, c->stream, n, P, out); break;
    EV_CASE(1) EV_CASE(2) EV_CASE(3) EV_CASE(4) EV_CASE(5) EV_CASE(6) EV_CASE(7) EV_CASE(8)
#undef EV_CASE
    }
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// debugging aid of the Foam layer (FFM_SYNC_RANGE): every queue of the device drained
extern "C" int ffm_device_synchronize(void) { FFM_HIP(hipDeviceSynchronize()); return FFM_OK; }
