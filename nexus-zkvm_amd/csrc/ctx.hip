// Context, device memory, staging ring, gather, timing.  Part of libnexus_hip.so (gfx950 only).
#include "internal.h"
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>

namespace nx {

thread_local std::string g_last_error;

int set_err(nx_ctx* ctx, int code, const std::string& msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}
int hip_fail(nx_ctx* ctx, hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return set_err(ctx, e == hipErrorOutOfMemory ? NX_ERR_OOM : NX_ERR_HIP, buf);
}

int stage(nx_ctx* ctx, const void* h_src, size_t bytes, void** d_out) {
    size_t need = (bytes + 255) & ~(size_t)255;
    if (need > ctx->scratch_size) return set_err(ctx, NX_ERR_ARG, "stage: request larger than scratch ring");
    if (ctx->scratch_off + need > ctx->scratch_size) {
        // wrap: make sure every earlier consumer of the ring has finished
        NX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        NX_HIP(ctx, hipStreamSynchronize(ctx->hash_stream));
        for (int i = 0; i < 3; i++) NX_HIP(ctx, hipStreamSynchronize(ctx->side[i]));
        ctx->scratch_off = 0;
    }
    memcpy(ctx->h_scratch + ctx->scratch_off, h_src, bytes);
    NX_HIP(ctx, hipMemcpyAsync(ctx->d_scratch + ctx->scratch_off, ctx->h_scratch + ctx->scratch_off, bytes,
                               hipMemcpyHostToDevice, ctx->stream));
    *d_out = ctx->d_scratch + ctx->scratch_off;
    ctx->scratch_off += need;
    return NX_OK;
}

constexpr size_t BOUNCE_HALF = (size_t)4 << 20, BOUNCE_MIN = (size_t)32 << 10;
static int bounce_ready(nx_ctx* ctx) {
    if (ctx->h_bounce) return NX_OK;
    NX_HIP(ctx, hipHostMalloc((void**)&ctx->h_bounce, 2 * BOUNCE_HALF, hipHostMallocDefault));
    for (int k = 0; k < 2; k++) NX_HIP(ctx, hipEventCreateWithFlags(&ctx->bounce_ev[k], hipEventDisableTiming));
    return NX_OK;
}
int copy_h2d_blocking(nx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t stream) {
    if (!bytes) return NX_OK;
    if (!stream) stream = ctx->stream;
    if (bytes < BOUNCE_MIN || host_pinned_by_owner(h_src, bytes)) {      // small: the runtime stages it by itself; pinned by its owner: DMA straight from it
        NX_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, stream));
        NX_HIP(ctx, hipStreamSynchronize(stream));
        return NX_OK;
    }
    NX_TRY(bounce_ready(ctx));
    size_t i = 0;
    for (size_t off = 0; off < bytes; off += BOUNCE_HALF, i++) {
        const int k = (int)(i & 1);
        const size_t n = std::min(BOUNCE_HALF, bytes - off);
        if (i >= 2) NX_HIP(ctx, hipEventSynchronize(ctx->bounce_ev[k]));          // the copy that read this half two chunks ago is done
        memcpy(ctx->h_bounce + k * BOUNCE_HALF, (const uint8_t*)h_src + off, n);
        NX_HIP(ctx, hipMemcpyAsync((uint8_t*)d_dst + off, ctx->h_bounce + k * BOUNCE_HALF, n, hipMemcpyHostToDevice, stream));
        NX_HIP(ctx, hipEventRecord(ctx->bounce_ev[k], stream));
    }
    NX_HIP(ctx, hipStreamSynchronize(stream));
    return NX_OK;
}
int copy_d2h_blocking(nx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!bytes) return NX_OK;
    if (bytes < BOUNCE_MIN || host_pinned_by_owner(h_dst, bytes)) {
        NX_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        NX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return NX_OK;
    }
    NX_TRY(bounce_ready(ctx));
    // chunk i lands in half i & 1 while chunk i - 1 is copied out to the caller
    const size_t n_chunks = (bytes + BOUNCE_HALF - 1) / BOUNCE_HALF;
    for (size_t i = 0; i <= n_chunks; i++) {
        if (i < n_chunks) {
            const int k = (int)(i & 1);
            const size_t off = i * BOUNCE_HALF, n = std::min(BOUNCE_HALF, bytes - off);
            NX_HIP(ctx, hipMemcpyAsync(ctx->h_bounce + k * BOUNCE_HALF, (const uint8_t*)d_src + off, n, hipMemcpyDeviceToHost, ctx->stream));
            NX_HIP(ctx, hipEventRecord(ctx->bounce_ev[k], ctx->stream));
        }
        if (i >= 1) {
            const int k = (int)((i - 1) & 1);
            const size_t off = (i - 1) * BOUNCE_HALF, n = std::min(BOUNCE_HALF, bytes - off);
            NX_HIP(ctx, hipEventSynchronize(ctx->bounce_ev[k]));
            memcpy((uint8_t*)h_dst + off, ctx->h_bounce + k * BOUNCE_HALF, n);
        }
    }
    return NX_OK;
}
int upload_async_staged(nx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    const size_t chunk = (size_t)4 << 20;
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = std::min(chunk, bytes - off);
        void* st = nullptr;
        NX_TRY(stage(ctx, (const uint8_t*)h_src + off, n, &st));
        NX_HIP(ctx, hipMemcpyAsync((uint8_t*)d_dst + off, st, n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return NX_OK;
}

int host_alloc(nx_ctx* ctx, size_t bytes, void** h_out) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = ctx->free_pinned.find(bytes);
    if (it != ctx->free_pinned.end()) { *h_out = it->second; ctx->free_pinned.erase(it); ctx->live_pinned[*h_out] = bytes; return NX_OK; }
    NX_HIP(ctx, hipHostMalloc(h_out, bytes, hipHostMallocDefault));
    ctx->live_pinned[*h_out] = bytes;
    return NX_OK;
}
void host_free(nx_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx->live_pinned.find(p);
    if (it == ctx->live_pinned.end()) return;
    ctx->free_pinned.insert({it->second, p});
    ctx->live_pinned.erase(it);
}

int make_colset(nx_ctx* ctx, const uint32_t* const* h_ptrs, uint32_t n, ColSet* out) {
    out->base = nullptr; out->stride = 0; out->table = nullptr;
    if (n == 0) return NX_OK;
    out->base = (uint32_t*)h_ptrs[0];
    if (n == 1) return NX_OK;
    bool uniform = h_ptrs[1] > h_ptrs[0];
    uint64_t stride = uniform ? (uint64_t)(h_ptrs[1] - h_ptrs[0]) : 0;
    for (uint32_t i = 2; uniform && i < n; i++)
        if (h_ptrs[i] != h_ptrs[0] + (uint64_t)i * stride) uniform = false;
    if (uniform) { out->stride = stride; return NX_OK; }
    void* d = nullptr;
    NX_TRY(stage(ctx, h_ptrs, (size_t)n * sizeof(uint32_t*), &d));
    out->table = (uint32_t* const*)d;
    return NX_OK;
}

int dev_alloc(nx_ctx* ctx, size_t bytes, void** out) {
    if (bytes == 0) bytes = 4;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = ctx->free_blocks.find(bytes);
    if (it != ctx->free_blocks.end()) {
        *out = it->second; ctx->free_blocks.erase(it); ctx->cached_bytes -= bytes;
        ctx->live_blocks[*out] = bytes;
        return NX_OK;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {  // give cached blocks back to the driver and retry once
        (void)hipGetLastError();
        dev_cache_release(ctx);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) { *out = nullptr; return hip_fail(ctx, e, "hipMalloc", __FILE__, __LINE__); }
    ctx->live_blocks[*out] = bytes;
    return NX_OK;
}
void dev_free(nx_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx->live_blocks.find(p);
    if (it == ctx->live_blocks.end()) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(p); return; }  // foreign pointer
    ctx->free_blocks.insert({it->second, p});
    ctx->cached_bytes += it->second;
    ctx->live_blocks.erase(it);
    if (ctx->cached_bytes > ((size_t)192 << 30)) dev_cache_release(ctx);  // keep the cache bounded (a 2^24-row prove recycles ~100 GB)
}
void dev_cache_release(nx_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->free_blocks) (void)hipFree(kv.second);
    ctx->free_blocks.clear(); ctx->cached_bytes = 0;
}

static hipEvent_t get_event(nx_ctx* ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
KTimer::KTimer(nx_ctx* c, int kind, uint64_t bytes, hipStream_t on_stream) : ctx(c), idx(-1), stream(on_stream ? on_stream : c->stream) {
    if (!c->timing) return;
    nx_ctx::Span s; s.e0 = get_event(c); s.e1 = get_event(c); s.kind = kind;
    (void)hipEventRecord(s.e0, stream);
    c->kind_bytes[kind] += bytes;
    idx = (int)c->spans.size();
    c->spans.push_back(s);
}
KTimer::~KTimer() { if (idx >= 0) (void)hipEventRecord(ctx->spans[idx].e1, stream); }
void timing_flush(nx_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->hash_stream);
    for (auto& s : ctx->spans) {
        float ms = 0; (void)hipEventElapsedTime(&ms, s.e0, s.e1);
        ctx->kind_ms[s.kind] += ms;
        ctx->event_pool.push_back(s.e0); ctx->event_pool.push_back(s.e1);
    }
    ctx->spans.clear();
}
void timing_reset(nx_ctx* ctx) {
    timing_flush(ctx);
    for (int i = 0; i < 4; i++) { ctx->kind_ms[i] = 0; ctx->kind_bytes[i] = 0; }
}

int streams_fork(nx_ctx* ctx, int n_streams) {
    if (n_streams <= 1) return NX_OK;
    NX_HIP(ctx, hipEventRecord(ctx->fork_ev, ctx->stream));
    for (int i = 0; i + 1 < n_streams && i < 3; i++) NX_HIP(ctx, hipStreamWaitEvent(ctx->side[i], ctx->fork_ev, 0));
    return NX_OK;
}
void streams_pick(nx_ctx* ctx, int batch_index, int n_streams) {
    int k = n_streams <= 1 ? 0 : batch_index % (n_streams > 4 ? 4 : n_streams);
    ctx->cur = k == 0 ? ctx->stream : ctx->side[k - 1];
}
int streams_join(nx_ctx* ctx, int n_streams) {
    ctx->cur = ctx->stream;
    if (n_streams <= 1) return NX_OK;
    for (int i = 0; i + 1 < n_streams && i < 3; i++) {
        NX_HIP(ctx, hipEventRecord(ctx->join_ev[i], ctx->side[i]));
        NX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->join_ev[i], 0));
    }
    return NX_OK;
}

// 16 bytes per lane, grid-stride: the plain streaming pattern (also the calibration kernel of tools/pmc_traffic.py)
__global__ __launch_bounds__(256) void copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void m31_add_into_kernel(u32* __restrict__ dst, const u32* __restrict__ src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = m_add(dst[i], src[i]);
}
__global__ __launch_bounds__(256) void m31_widen_kernel(u64* __restrict__ dst, const u32* __restrict__ src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void m31_narrow_kernel(u32* __restrict__ dst, const u64* __restrict__ src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (u32)(src[i] % P);
}

__global__ void gather_kernel(const uint32_t* const* ptrs, const uint64_t* index, size_t n, uint32_t* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = ptrs[i] ? ptrs[i][index[i]] : 0u;   // NULL: a word another GPU of a row-sharded prove supplies
}

// Column shard <-> row block transposition around an all-to-all.  `full` holds n_cols columns of `rows` words (column stride col_stride);
// `blocks` is [world][n_cols][rows / world]: block d = rows [d * rows/world, (d+1) * rows/world) of every column, i.e. what GPU d
// receives (pack) or what came from GPU d (unpack).  16 bytes per lane; rows / world is a multiple of 4.
__global__ __launch_bounds__(256) void transpose_blocks_kernel(uint32_t* full, uint64_t col_stride, uint32_t* blocks, uint32_t n_cols, uint64_t rows, uint32_t world, int unpack) {
    const uint64_t rb4 = rows / world / 4, total = (uint64_t)n_cols * world * rb4;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = i % rb4, cd = i / rb4;
        const uint32_t c = (uint32_t)(cd % n_cols), d = (uint32_t)(cd / n_cols);
        uint4* f = reinterpret_cast<uint4*>(full + (uint64_t)c * col_stride) + (uint64_t)d * rb4 + x;
        uint4* b = reinterpret_cast<uint4*>(blocks) + ((uint64_t)d * n_cols + c) * rb4 + x;
        if (unpack) *f = *b; else *b = *f;
    }
}
int transpose_blocks(nx_ctx* ctx, uint32_t* full, uint64_t col_stride, uint32_t* blocks, uint32_t n_cols, uint64_t rows, uint32_t world, bool unpack) {
    if (!n_cols || !rows) return NX_OK;
    if ((rows / world) % 4 || rows % world || col_stride % 4) return set_err(ctx, NX_ERR_ARG, "transpose_blocks: the row block must be a multiple of 4 words");
    const uint64_t total = (uint64_t)n_cols * rows / 4;
    hipLaunchKernelGGL(transpose_blocks_kernel, dim3((unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)ctx->n_cus * 32)), dim3(256), 0, ctx->stream, full, col_stride, blocks, n_cols,
                       rows, world, unpack ? 1 : 0);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

}  // namespace nx

using namespace nx;

// Column::clone of many equally long columns in ONE launch (a wide AIR keeps ~10^3 main columns per component for its logup
// fractions: one nx_copy each was ~2.5 us of launch per 16..64 KB of work).  words % 4 == 0, 16-byte aligned columns.
__global__ __launch_bounds__(256) void copy_cols_kernel(nx::ColSet dst, nx::ColSet src, size_t n4) {
    const uint4* __restrict__ s = (const uint4*)src.col(blockIdx.y);
    uint4* __restrict__ d = (uint4*)dst.col(blockIdx.y);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
namespace nx {
int copy_columns(nx_ctx* ctx, uint32_t* const* h_dst, const uint32_t* const* h_src, uint32_t n_cols, size_t n_words) {
    if (n_cols == 0 || n_words == 0) return NX_OK;
    bool vec = (n_words & 3) == 0;
    for (uint32_t k = 0; k < n_cols && vec; k++) vec = !(((uintptr_t)h_dst[k] | (uintptr_t)h_src[k]) & 15);
    if (!vec || n_cols < 4) { for (uint32_t k = 0; k < n_cols; k++) NX_TRY(nx_copy(ctx, h_dst[k], h_src[k], n_words)); return NX_OK; }
    for (uint32_t c0 = 0; c0 < n_cols; c0 += 65535) {
        const uint32_t nb = std::min<uint32_t>(65535, n_cols - c0);
        ColSet d, s;
        NX_TRY(make_colset(ctx, h_dst + c0, nb, &d));
        NX_TRY(make_colset(ctx, (uint32_t* const*)(h_src + c0), nb, &s));
        const size_t n4 = n_words / 4;
        const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((n4 + 255) / 256, std::max<size_t>(1, (size_t)ctx->n_cus * 16 / nb)));
        hipLaunchKernelGGL(copy_cols_kernel, dim3(bx, nb), dim3(256), 0, ctx->stream, d, s, n4);
        NX_LAUNCH_CHECK(ctx);
    }
    return NX_OK;
}
}  // namespace nx

extern "C" {

const char* nx_version(void) { return "nexus_hip 0.3 (gfx950)"; }

const char* nx_last_error(const nx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static nx_options options_from_env() {
    nx_options o;
    o.fft_batch_cols = std::max(1, env_int("NX_FFT_BATCH", 2));
    o.fft_streams = clampi(env_int("NX_FFT_STREAMS", 2), 1, 4);
    o.fri_dist_min_log = std::max(0, env_int("NX_FRI_DIST_MIN_LOG", 21));
    o.dist_chunks = std::max(0, env_int("NX_DIST_CHUNKS", 0));
    o.air_segment = std::max(200, env_int("NX_AIR_SEGMENT", 9000));
    o.air_degree_split = env_int("NX_AIR_DEGREE_SPLIT", 1) != 0;
    o.quotients_coeffs = env_int("NX_QUOTIENTS_COEFFS", 1) != 0;
    o.air_half_domain = env_int("NX_AIR_HALF_DOMAIN", 1) != 0;
    o.air_quarter_domain = std::max(0, std::min(2, env_int("NX_AIR_QUARTER_DOMAIN", 2)));   // 2: also constraints that read a neighbour row
    o.comm_timeout_ms = std::max(0, env_int("NX_COMM_TIMEOUT_MS", 120000));
    o.fft_kmax = clampi(env_int("NX_FFT_KMAX", 9), 1, 11);
    o.fft_fused = env_int("NX_FFT_FUSED", 1) != 0;
    o.merkle_fused = clampi(env_int("NX_MERKLE_FUSED", 21), 0, 31);   // profiles/r06_merkle_fused_ab.jsonl: FRI stage 3.48 -> 3.37 ms at 21, the prove within its noise
    o.merkle_subtree = clampi(env_int("NX_MERKLE_SUBTREE", 14), 0, 30);   // profiles/r06_merkle_top_ab.jsonl: top 7 / subtree 14 against 10 / 17
    o.merkle_top = clampi(env_int("NX_MERKLE_TOP", 7), 1, 10);
    o.merkle_pair_levels = env_int("NX_MERKLE_PAIR_LEVELS", 1) != 0;
    { int x = env_int("NX_PIPE_COLS", 0); o.commit_pipe_cols = x < 16 ? 0 : (x / 16) * 16; }
    o.fri_device_channel = env_int("NX_FRI_DEVICE_CHANNEL", 1) != 0;
    o.fri_tail = clampi(env_int("NX_FRI_TAIL", 1), 0, 11);
    o.logup_scan_tiled = env_int("NX_LOGUP_SCAN_TILED", 1) != 0;
    o.logup_staged = env_int("NX_LOGUP_STAGED", 1) != 0;
    o.logup_per_column = env_int("NX_LOGUP_PER_COLUMN", 0) != 0;
    o.machine_reuse_pre = env_int("NX_MACHINE_REUSE_PREPROCESSED", 0) != 0;
    o.machine_logup_program = env_int("NX_MACHINE_LOGUP_PROGRAM", 0) != 0;
    o.machine_queue_trees = env_int("NX_MACHINE_QUEUE_TREES", 0) != 0;   // measured: no gain (profiles/r05_queue_trees_ab.txt)
    return o;
}
struct OptEntry { const char* name; int nx_options::*field; int lo, hi; };
static const OptEntry k_options[] = {
    {"fft.batch_cols", &nx_options::fft_batch_cols, 1, 256},
    {"fft.streams", &nx_options::fft_streams, 1, 4},
    {"fri.dist_min_log", &nx_options::fri_dist_min_log, 0, 31},
    {"dist.chunks", &nx_options::dist_chunks, 0, 64},
    {"air.segment", &nx_options::air_segment, 200, 1 << 30},
    {"air.degree_split", &nx_options::air_degree_split, 0, 1},
    {"quotients.coeffs", &nx_options::quotients_coeffs, 0, 1},
    {"air.half_domain", &nx_options::air_half_domain, 0, 1},
    {"air.quarter_domain", &nx_options::air_quarter_domain, 0, 2},
    {"comm.timeout_ms", &nx_options::comm_timeout_ms, 0, 1 << 30},
    {"fft.kmax", &nx_options::fft_kmax, 1, 11},
    {"fft.fused", &nx_options::fft_fused, 0, 1},
    {"merkle.fused", &nx_options::merkle_fused, 0, 31},
    {"merkle.subtree", &nx_options::merkle_subtree, 0, 30},
    {"merkle.top", &nx_options::merkle_top, 1, 10},
    {"merkle.pair_levels", &nx_options::merkle_pair_levels, 0, 1},
    {"commit.pipe_cols", &nx_options::commit_pipe_cols, 0, 1 << 20},
    {"fri.device_channel", &nx_options::fri_device_channel, 0, 1},
    {"fri.tail", &nx_options::fri_tail, 0, 11},
    {"logup.scan_tiled", &nx_options::logup_scan_tiled, 0, 1},
    {"logup.staged", &nx_options::logup_staged, 0, 1},
    {"logup.per_column", &nx_options::logup_per_column, 0, 1},
    {"machine.reuse_preprocessed", &nx_options::machine_reuse_pre, 0, 1},
    {"machine.queue_trees", &nx_options::machine_queue_trees, 0, 1},
    {"machine.logup_program", &nx_options::machine_logup_program, 0, 1},
};
int nx_ctx_set_option(nx_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return set_err(ctx, NX_ERR_ARG, "nx_ctx_set_option: NULL argument");
    for (const OptEntry& e : k_options)
        if (!strcmp(e.name, name)) {
            if (value < e.lo || value > e.hi) return set_err(ctx, NX_ERR_ARG, std::string("nx_ctx_set_option: value out of range for ") + name);
            ctx->opt.*(e.field) = (int)value;
            if (e.field == &nx_options::commit_pipe_cols) ctx->opt.commit_pipe_cols = value < 16 ? 0 : (int)(value / 16) * 16;   // whole 16-column hash blocks, like NX_PIPE_COLS
            return NX_OK;
        }
    return set_err(ctx, NX_ERR_ARG, std::string("nx_ctx_set_option: unknown option ") + name);
}
int nx_ctx_get_option(const nx_ctx* ctx, const char* name, int64_t* value) {
    if (!ctx || !name || !value) return set_err(const_cast<nx_ctx*>(ctx), NX_ERR_ARG, "nx_ctx_get_option: NULL argument");
    for (const OptEntry& e : k_options)
        if (!strcmp(e.name, name)) { *value = ctx->opt.*(e.field); return NX_OK; }
    return set_err(const_cast<nx_ctx*>(ctx), NX_ERR_ARG, std::string("nx_ctx_get_option: unknown option ") + name);
}

// NX_ABORT_BACKTRACE=1: the native stack of whoever calls abort() (a runtime assertion, a GPU fault handler, glibc's heap checks) on stderr
static void abort_backtrace(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "\n[nexus_hip] SIGABRT — native stack of the aborting thread:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
int nx_ctx_create(int device, nx_ctx** out) {
    if (!out) return set_err(nullptr, NX_ERR_ARG, "nx_ctx_create: out is NULL");
    { static bool once = false; if (!once) { once = true; const char* e = getenv("NX_ABORT_BACKTRACE"); if (e && *e == '1') signal(SIGABRT, abort_backtrace); } }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_err(nullptr, NX_ERR_NO_DEVICE, "nx_ctx_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= count) return set_err(nullptr, NX_ERR_ARG, "nx_ctx_create: bad device index");
    hipDeviceProp_t prop;
    NX_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return set_err(nullptr, NX_ERR_NO_DEVICE, std::string("nx_ctx_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    NX_HIP(nullptr, hipSetDevice(device));
    nx_ctx* c = new nx_ctx();
    c->opt = options_from_env();
    c->device = device; c->hash_mode = NX_HASH_BLAKE2S; c->timing = false; c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->scratch_size = 16u << 20; c->scratch_off = 0; c->d_scratch = nullptr; c->h_scratch = nullptr; c->cached_bytes = 0;
    for (int i = 0; i < 4; i++) { c->kind_ms[i] = 0; c->kind_bytes[i] = 0; }
    NX_HIP(nullptr, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->cur = c->stream;
    NX_HIP(nullptr, hipStreamCreateWithFlags(&c->hash_stream, hipStreamNonBlocking));
    NX_HIP(nullptr, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    NX_HIP(nullptr, hipStreamCreateWithFlags(&c->perm_stream, hipStreamNonBlocking));
    NX_HIP(nullptr, hipEventCreateWithFlags(&c->hash_ev, hipEventDisableTiming));
    NX_HIP(nullptr, hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
    for (int i = 0; i < 3; i++) {
        NX_HIP(nullptr, hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        NX_HIP(nullptr, hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming));
    }
    NX_HIP(nullptr, hipMalloc((void**)&c->d_scratch, c->scratch_size));
    NX_HIP(nullptr, hipHostMalloc((void**)&c->h_scratch, c->scratch_size, hipHostMallocDefault));
    *out = c;
    return NX_OK;
}

void nx_ctx_destroy(nx_ctx* ctx) {
    NX_GUARD(ctx);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    nxhip::machine_kernels_release(ctx);
    nx::logup_kernels_release(ctx);
    timing_flush(ctx);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    dev_cache_release(ctx);
    for (auto& kv : ctx->live_blocks) (void)hipFree(kv.first);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
    if (ctx->h_bounce) { (void)hipHostFree(ctx->h_bounce); for (int k = 0; k < 2; k++) (void)hipEventDestroy(ctx->bounce_ev[k]); }
    for (auto& kv : ctx->free_pinned) (void)hipHostFree(kv.second);
    for (auto& kv : ctx->live_pinned) (void)hipHostFree(kv.first);
    for (int i = 0; i < 3; i++) { (void)hipStreamDestroy(ctx->side[i]); (void)hipEventDestroy(ctx->join_ev[i]); }
    (void)hipEventDestroy(ctx->fork_ev);
    (void)hipStreamDestroy(ctx->hash_stream); (void)hipEventDestroy(ctx->hash_ev);
    (void)hipStreamDestroy(ctx->copy_stream); (void)hipStreamDestroy(ctx->perm_stream);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int nx_ctx_set_hash_mode(nx_ctx* ctx, int mode) {
    if (!ctx) return set_err(nullptr, NX_ERR_ARG, "nx_ctx_set_hash_mode: NULL context");
    if (mode != NX_HASH_BLAKE2S && mode != NX_HASH_BLAKE2S_RAW0) return set_err(ctx, NX_ERR_ARG, "bad hash mode");
    ctx->hash_mode = mode;
    return NX_OK;
}

int nx_sync(nx_ctx* ctx) { NX_GUARD(ctx); if (!ctx) return set_err(nullptr, NX_ERR_ARG, "nx_sync: NULL context"); NX_HIP(ctx, hipStreamSynchronize(ctx->stream)); NX_HIP(ctx, hipStreamSynchronize(ctx->hash_stream)); return NX_OK; }
int nx_ctx_trim(nx_ctx* ctx) { NX_GUARD(ctx); if (!ctx) return set_err(nullptr, NX_ERR_ARG, "nx_ctx_trim: NULL context"); NX_TRY(nx_sync(ctx)); dev_cache_release(ctx); return NX_OK; }
void* nx_ctx_stream(nx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int nx_alloc(nx_ctx* ctx, size_t n_words, uint32_t** d_out) {
    NX_GUARD(ctx);
    if (!ctx || !d_out) return set_err(ctx, NX_ERR_ARG, "nx_alloc: NULL argument");
    *d_out = nullptr;
    return dev_alloc(ctx, n_words * 4, (void**)d_out);
}
int nx_free(nx_ctx* ctx, uint32_t* d_ptr) {
    NX_GUARD(ctx);
    if (!ctx) return set_err(nullptr, NX_ERR_ARG, "nx_free: NULL context");
    dev_free(ctx, d_ptr);
    return NX_OK;
}
int nx_memset_zero(nx_ctx* ctx, uint32_t* d_ptr, size_t n_words) {
    NX_GUARD(ctx);
    if (!ctx || (!d_ptr && n_words)) return set_err(ctx, NX_ERR_ARG, "nx_memset_zero: NULL argument");
    if (!n_words) return NX_OK;
    NX_HIP(ctx, hipMemsetAsync(d_ptr, 0, n_words * 4, ctx->stream));
    return NX_OK;
}
int nx_upload(nx_ctx* ctx, uint32_t* d_dst, const uint32_t* h_src, size_t n_words) {
    NX_GUARD(ctx);
    if (!ctx || ((!d_dst || !h_src) && n_words)) return set_err(ctx, NX_ERR_ARG, "nx_upload: NULL argument");
    if (!n_words) return NX_OK;
    return copy_h2d_blocking(ctx, d_dst, h_src, n_words * 4);   // complete on return; the caller's (possibly pageable) pages are never pinned in place
}
int nx_download(nx_ctx* ctx, uint32_t* h_dst, const uint32_t* d_src, size_t n_words) {
    NX_GUARD(ctx);
    if (!ctx || ((!h_dst || !d_src) && n_words)) return set_err(ctx, NX_ERR_ARG, "nx_download: NULL argument");
    if (!n_words) return NX_OK;
    return copy_d2h_blocking(ctx, h_dst, d_src, n_words * 4);
}

int nx_copy(nx_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t n_words) {
    NX_GUARD(ctx);
    if (!ctx || ((!d_dst || !d_src) && n_words)) return set_err(ctx, NX_ERR_ARG, "nx_copy: NULL argument");
    if (n_words == 0) return NX_OK;
    if ((n_words & 3) || ((uintptr_t)d_dst & 15) || ((uintptr_t)d_src & 15)) {
        NX_HIP(ctx, hipMemcpyAsync(d_dst, d_src, n_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return NX_OK;
    }
    size_t n4 = n_words / 4;
    unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, (size_t)ctx->n_cus * 16);
    hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (uint4*)d_dst, (const uint4*)d_src, n4);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

static unsigned stream_grid(nx_ctx* ctx, size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx->n_cus * 16); }
int nx_m31_add_into(nx_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t n_words) {
    NX_GUARD(ctx);
    if (!n_words) return NX_OK;
    hipLaunchKernelGGL(m31_add_into_kernel, dim3(stream_grid(ctx, n_words)), dim3(256), 0, ctx->stream, d_dst, d_src, n_words);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}
int nx_m31_widen(nx_ctx* ctx, uint64_t* d_dst, const uint32_t* d_src, size_t n_words) {
    NX_GUARD(ctx);
    if (!n_words) return NX_OK;
    hipLaunchKernelGGL(m31_widen_kernel, dim3(stream_grid(ctx, n_words)), dim3(256), 0, ctx->stream, (u64*)d_dst, d_src, n_words);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}
int nx_m31_narrow(nx_ctx* ctx, uint32_t* d_dst, const uint64_t* d_src, size_t n_words) {
    NX_GUARD(ctx);
    if (!n_words) return NX_OK;
    hipLaunchKernelGGL(m31_narrow_kernel, dim3(stream_grid(ctx, n_words)), dim3(256), 0, ctx->stream, d_dst, (const u64*)d_src, n_words);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int nx_gather(nx_ctx* ctx, const uint32_t* const* d_ptrs, const uint64_t* index, size_t n, uint32_t* h_out) {
    NX_GUARD(ctx);
    if (!ctx || ((!d_ptrs || !index || !h_out) && n)) return set_err(ctx, NX_ERR_ARG, "nx_gather: NULL argument");
    if (n == 0) return NX_OK;
    uint8_t* d = nullptr;
    size_t bytes = n * (8 + 8 + 4);
    NX_TRY(dev_alloc(ctx, bytes, (void**)&d));
    const uint32_t** dp = (const uint32_t**)d;
    uint64_t* di = (uint64_t*)(d + n * 8);
    uint32_t* dout = (uint32_t*)(d + n * 16);
    hipError_t e = hipSuccess;
    if (upload_async_staged(ctx, dp, d_ptrs, n * 8) != NX_OK || upload_async_staged(ctx, di, index, n * 8) != NX_OK) e = hipErrorUnknown;   // the caller's arrays are never pinned in place (internal.h, h_bounce)
    if (e == hipSuccess) {
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dp, di, n, dout);
        e = hipGetLastError();
    }
    int rc = NX_OK;
    if (e == hipSuccess) rc = copy_d2h_blocking(ctx, h_out, dout, n * 4);
    else (void)hipStreamSynchronize(ctx->stream);
    dev_free(ctx, d);
    if (e != hipSuccess) return hip_fail(ctx, e, "nx_gather", __FILE__, __LINE__);
    return rc;
}

void nx_free_host(void* p) { free(p); }

}  // extern "C"
