// Host-side staging of a block-Toeplitz dictionary (JSTSP_HOST calls of proposed_algorithm; round 5).
//
// The dictionaries the reference's drivers build stack, for each delay ld = 0..L-1, the pilot frame delayed by ld samples
// under the Gt transmit steering vectors (proposed_hbf.m:17,36-42; plot_errorVSsnr.m:132-136), so that
// B(ld Gt + g, m) == B(g, m - ld) for m >= ld, bit for bit: 7/8 of the 16 MiB a trial's dictionary occupies at BASELINE
// configs[1] repeats its first block.  The device path already PROBES that (fused.hip) - after the whole array has crossed
// the link.  Here the test runs on the host, in the same loop that copies what is needed into the pinned staging buffer:
// a few threads walk the caller's array once (it has to be read anyway), compare every delayed block with the shifted
// first block (memcmp: bit-wise, the stricter test) and copy only block 0 and the `ld` leading columns of each block ld;
// the link carries 1/L of the dictionary and `expand_toeplitz_kernel` rebuilds the full array in HBM - the SAME bits the
// caller holds, so everything downstream (packing, Grams, the recovery path) is unchanged and the results are bit-identical.
// The first mismatch ends the attempt and the array is uploaded as it is.  jstsp_c64 (MATLAB's doubles) take the same
// route with the narrowing to fp32 folded into the copy: the link carries floats.
#include "common.h"
#include "solver_common.h"

#include <atomic>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

namespace jstsp {
namespace {

int host_threads(int work_items)
{
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 4;
    n = std::max(1, n / 2);                                   // SMT siblings add no memory bandwidth
    std::ifstream f("/sys/fs/cgroup/cpu.max");                // cgroup v2 CPU quota of this process, if any
    std::string q;
    long long per = 0;
    if (f >> q >> per && q != "max" && per > 0) n = std::min<long long>(n, std::max<long long>(1, std::atoll(q.c_str()) / per));
    if (const char *e = getenv("JSTSP_HOST_THREADS")) n = std::max(1, atoi(e));
    return std::max(1, std::min(std::min(n, 16), work_items));
}

inline void copy_block(float2 *dst, const float2 *src, int n) { memcpy(dst, src, (size_t)n * sizeof(float2)); }
inline void copy_block(float2 *dst, const double2 *src, int n)
{
    for (int i = 0; i < n; ++i) dst[i] = make_float2((float)src[i].x, (float)src[i].y);
}

// One dictionary (G2 x M column-major, ld = G2): true if block-Toeplitz with block height gt.  blk0 (gt x M) and lead (the ld
// leading columns of block ld, ld = 1..L-1, gt entries each, ordered by (ld, m)) receive the compact form when non-NULL.
template <class T> bool check_one(const T *B, int G2, int M, int gt, float2 *blk0, float2 *lead, const std::atomic<int> *stop)
{
    const int L = G2 / gt;
    const size_t bytes = (size_t)gt * sizeof(T);
    for (int m = 0; m < M; ++m) {
        if (stop && (m & 63) == 0 && stop->load(std::memory_order_relaxed)) return false;
        const T *col = B + (size_t)m * G2;
        for (int ld = 1; ld < L; ++ld) {
            if (m >= ld) {
                if (memcmp(col + (size_t)ld * gt, B + (size_t)(m - ld) * G2, bytes) != 0) return false;
            } else if (lead) {
                copy_block(lead + ((size_t)ld * (ld - 1) / 2 + m) * gt, col + (size_t)ld * gt, gt);
            }
        }
        if (blk0) copy_block(blk0 + (size_t)m * gt, col, gt);
    }
    return true;
}

// B[t][ld gt + g + G2 m] = m >= ld ? blk0[t][g + gt (m - ld)] : lead[t][(ld (ld - 1) / 2 + m) gt + g]
__global__ __launch_bounds__(256) void expand_toeplitz_kernel(const float2 *__restrict__ C, long long sC, int G2, int M, int gt, int gsh,
                                                              float2 *__restrict__ B, long long sB)
{
    const int t = blockIdx.y;
    const float2 *blk0 = C + (long long)t * sC, *lead = blk0 + (long long)gt * M;
    float2 *out = B + (long long)t * sB;
    const long long n = (long long)G2 * M;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e % G2), m = (int)(e / G2);
        const int ld = r >> gsh, g = r & (gt - 1);
        out[e] = (m >= ld) ? blk0[g + (long long)gt * (m - ld)] : lead[((long long)ld * (ld - 1) / 2 + m) * gt + g];
    }
}

}  // namespace

// Stage `nB` dictionaries (G2 x M each, contiguous: stride G2 M) from host memory into Bdev (device, nB G2 M float2).
// *gt_out = block height when the compact route was taken (Bdev then holds the expanded array, enqueued on ctx->stream),
// 0 when the structure is absent - nothing has been uploaded then and the caller copies the array as it is.
template <class T>
int host_toeplitz_stage(jstsp_ctx *ctx, const T *Bh, int G2, int M, int nB, float2 *Bdev, float2 *Cdev, size_t cdev_elems, int *gt_out)
{
    *gt_out = 0;
    const size_t per = (size_t)G2 * M;
    // candidate block heights on the first dictionary (a wrong one fails within the first columns)
    int gt = 0;
    for (int c = 16; c <= 256 && !gt; c *= 2)
        if (G2 % c == 0 && 2 * c <= G2 && M >= G2 / c && check_one(Bh, G2, M, c, nullptr, nullptr, nullptr)) gt = c;
    if (!gt) return 0;
    const int L = G2 / gt;
    const size_t nlead = (size_t)L * (L - 1) / 2 * gt, cper = (size_t)gt * M + nlead;
    if (cdev_elems < cper * nB) return 0;
    // the pinned staging buffer of the context (grow-only); the previous call's copy out of it must have completed
    const size_t need = cper * nB * sizeof(float2);
    if (ctx->hpin_pending) { JSTSP_HIP(hipEventSynchronize(ctx->hpin_done)); ctx->hpin_pending = false; }
    if (ctx->hpin_cap < need) {
        if (ctx->hpin) (void)hipHostFree(ctx->hpin);
        ctx->hpin = nullptr; ctx->hpin_cap = 0;
        JSTSP_HIP(hipHostMalloc(&ctx->hpin, need + need / 8, hipHostMallocDefault));
        ctx->hpin_cap = need + need / 8;
    }
    if (!ctx->hpin_done) JSTSP_HIP(hipEventCreateWithFlags(&ctx->hpin_done, hipEventDisableTiming));
    float2 *pin = reinterpret_cast<float2 *>(ctx->hpin);
    std::atomic<int> stop{0};
    const int nth = host_threads(nB);
    auto work = [&](int k) {
        for (int t = k; t < nB; t += nth) {
            float2 *c = pin + (size_t)t * cper;
            if (!check_one(Bh + (size_t)t * per, G2, M, gt, c, c + (size_t)gt * M, &stop)) { stop.store(1); return; }
        }
    };
    {
        // (thread creation can throw - std::system_error at a thread / pid limit of the cgroup - and this is below an extern "C"
        //  entry: the stripes whose thread did not start are worked off here, nothing propagates)
        std::vector<std::thread> th;
        std::vector<int> mine{0};
        for (int k = 1; k < nth; ++k) {
            try { th.emplace_back(work, k); } catch (...) { mine.push_back(k); }
        }
        for (int k : mine) work(k);
        for (auto &x : th) x.join();
    }
    if (stop.load()) return 0;                                // one dictionary without the structure: the plain upload
    JSTSP_HIP(hipMemcpyAsync(Cdev, pin, need, hipMemcpyHostToDevice, ctx->stream));
    JSTSP_HIP(hipEventRecord(ctx->hpin_done, ctx->stream));
    ctx->hpin_pending = true;
    hipLaunchKernelGGL(expand_toeplitz_kernel, dim3(2048, nB), dim3(256), 0, ctx->stream, Cdev, (long long)cper, G2, M, gt,
                       31 - __builtin_clz((unsigned)gt), Bdev, (long long)per);
    JSTSP_HIP(hipGetLastError());
    *gt_out = gt;
    return 0;
}

size_t host_toeplitz_compact_elems(int G2, int M, int nB)
{
    // room for the compact form of block heights up to 64 (a larger block height found on the host takes the plain upload):
    // gt M entries of block 0 + L (L - 1) / 2 gt <= G2 (G2 / 16) / 2 leading entries per dictionary
    const int gt = std::min(64, G2 / 2);
    return ((size_t)gt * M + (size_t)G2 * (G2 / 16) / 2 + 64) * nB;
}

template int host_toeplitz_stage<float2>(jstsp_ctx *, const float2 *, int, int, int, float2 *, float2 *, size_t, int *);
template int host_toeplitz_stage<double2>(jstsp_ctx *, const double2 *, int, int, int, float2 *, float2 *, size_t, int *);

}  // namespace jstsp
