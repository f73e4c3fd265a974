// nf_locator.h -- the locator shared by the weight build (nf_weights.hip) and the point location (nf_vinterp.hip): the box
// hierarchy over the cells of a grid, the breadth-first walk over it, and the scratch memory of a build.  Internal header.
#pragma once
#include <cstring>  // rocprim's texture iterator needs host memset declared first
#include <rocprim/device/device_scan.hpp>
#include <vector>

#include "nf_common.h"

namespace nf {

// ---- the locator: a box hierarchy over the cells, walked breadth-first by all segment images at once ----------------------
// mint's buildLocator bins the cells into buckets (field.py:47: numCellsPerBucket = 128) and computeWeights asks the buckets a
// target segment passes through.  Here: the cells are grouped 16 by 16 by 16 ... (16 cells, 256, 4096, ... up to one root),
// every group with the bounding box of its cells: 4 x 4 blocks when the row length of the grid is known (struct Layout), 16
// consecutive cells otherwise (on a row-major grid: strips of a grid row, then bundles of rows; any order gives valid boxes, a
// coherent one gives small boxes).
// All (group, segment image) pairs whose box the segment may touch are expanded level by level, ONE LANE PER (pair, child):
// bounding boxes first, then -- no divisions -- on which side of the target line the box's corners lie.  Every level is a count
// pass (ballot masks per wavefront), a scan and a fill pass, so the pairs of a level come out in (image, group) order, run to
// run the same; the last level tests the boxes of the cells and leaves the (cell, image) candidates of the clip stage.
// The work follows the number of cells the lines cross, not cells x segments: profiles/r05_weights_scaling.txt.
struct Box4 {
    float xmin, xmax, ymin, ymax;    // expanded by the slack of the cells inside, then rounded OUTWARD to float (a box only has to
                                     // hold its cells: 16 bytes per box instead of 32 halve the walk's traffic); xmin > xmax:
                                     // nothing inside
};
__device__ inline Box4 outward(double xmin, double xmax, double ymin, double ymax)
{
    return Box4{__double2float_rd(xmin), __double2float_ru(xmax), __double2float_rd(ymin), __double2float_ru(ymax)};
}
constexpr int kFan = 16;

// one cell's unwrapped corners, its bounding box and the slack the tests give it; false: not a cell (non-finite corner)
__device__ inline bool cell_geometry(const double *__restrict__ xy, long c, double period, double *v, double &cxmin,
                                     double &cxmax, double &cymin, double &cymax, double &slack)
{
    const double2 *p = reinterpret_cast<const double2 *>(xy + 8 * c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double2 t = p[k];
        v[2 * k] = t.x;
        v[2 * k + 1] = t.y;
    }
    cxmin = 1e300, cxmax = -1e300, cymin = 1e300, cymax = -1e300, slack = 0.0;
    if (!quad_is_finite(v)) return false;   // NaN / infinite corners: not a cell
    unwrap_quad(v, period);                  // date-line cells (nf_common.h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cxmin = fmin(cxmin, v[2 * i]);
        cxmax = fmax(cxmax, v[2 * i]);
        cymin = fmin(cymin, v[2 * i + 1]);
        cymax = fmax(cymax, v[2 * i + 1]);
    }
    slack = 1.e-9 * (fabs(cxmin) + fabs(cxmax) + fabs(cymin) + fabs(cymax) + 1.0);
    return true;
}

// box of kFan consecutive lanes' boxes -> the group's first lane
__device__ inline Box4 fan_union(double xmin, double xmax, double ymin, double ymax)
{
#pragma unroll
    for (int o = kFan / 2; o > 0; o >>= 1) {
        xmin = fmin(xmin, __shfl_xor(xmin, o, kFan));
        xmax = fmax(xmax, __shfl_xor(xmax, o, kFan));
        ymin = fmin(ymin, __shfl_xor(ymin, o, kFan));
        ymax = fmax(ymax, __shfl_xor(ymax, o, kFan));
    }
    return outward(xmin, xmax, ymin, ymax);
}

// How the nodes of one level are laid out and which 16 nodes of the level below a node holds.  The cells of a Field are a
// (ny, nx) array: its groups are 4 x 4 blocks of cells, then 4 x 4 blocks of blocks ... -- compact boxes, so that the number
// of (group, line) pairs falls by four from one level to the next coarser one.  A mint.Grid is a flat list of cells whose row
// length nobody told us: one row of nodes, 16 consecutive ones per group (on a row-major grid: strips of a grid row).
struct Layout {
    int ty, tx;     // the level below: ty rows of tx nodes (cells at the last level)
    int fy, fx;     // a node holds fy x fx of them (fy * fx = 16)
    int px;         // nodes per row of THIS level
};
// child j (0 .. 15) of node p, or -1 when it lies beyond the edge of the level below
__device__ inline long child_of(const Layout &g, long p, int j)
{
    const long py = p / g.px, pxx = p - py * g.px;
    const long cy = py * g.fy + j / g.fx, cx = pxx * g.fx + j % g.fx;
    return (cy < g.ty && cx < g.tx) ? cy * g.tx + cx : -1;
}

// level 0: the box of every cell, from the corner table
static __global__ __launch_bounds__(kBlock) void k_boxes_cells(const double *__restrict__ xy, long ncell, double period,
                                                        Box4 *__restrict__ box0)
{
    const long c = (long)blockIdx.x * kBlock + threadIdx.x;
    if (c >= ncell) return;
    double v[8], cxmin, cxmax, cymin, cymax, slack;
    cell_geometry(xy, c, period, v, cxmin, cxmax, cymin, cymax, slack);
    // a non-finite cell has an empty box and no slack: it adds nothing to its group
    box0[c] = outward(cxmin - slack, cxmax + slack, cymin - slack, cymax + slack);
}

// level l + 1 from level l: lane j of every 16 loads child j of its node
static __global__ __launch_bounds__(kBlock) void k_boxes_up(const Box4 *__restrict__ in, Layout g, long n_out, Box4 *__restrict__ out)
{
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    const long p = t / kFan;
    Box4 b = outward(1e300, -1e300, 1e300, -1e300);
    if (p < n_out) {
        const long c = child_of(g, p, (int)(t & (kFan - 1)));
        if (c >= 0) b = in[c];
    }
    b = fan_union(b.xmin, b.xmax, b.ymin, b.ymax);     // floats are doubles: nothing moves
    if ((threadIdx.x & (kFan - 1)) == 0 && p < n_out) out[p] = b;
}

// count pass of a count / scan / fill step: every wavefront leaves its ballot mask, every workgroup the number of set bits
// of its four masks (the scan then runs over workgroups: a quarter of the entries)
__device__ inline void block_count(unsigned long long mask, unsigned long long *__restrict__ wmask, int *__restrict__ bcnt)
{
    __shared__ int s_c[kBlock / kWave];
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        wmask[(long)blockIdx.x * (kBlock / kWave) + w] = mask;
        s_c[w] = __popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
#pragma unroll
        for (int k = 0; k < kBlock / kWave; ++k) n += s_c[k];
        bcnt[blockIdx.x] = n;
    }
}
// fill pass: where the set bits of wavefront w of a workgroup start = the workgroup's offset + the bits of the waves before it
__device__ inline long wave_offset(const unsigned long long *__restrict__ wmask, const long *__restrict__ boff, long block, int w)
{
    long pos = boff[block];
    for (int k = 0; k < w; ++k) pos += __popcll(wmask[block * (kBlock / kWave) + k]);
    return pos;
}

// fill pass: the children that passed, in (pair, child) order
static __global__ __launch_bounds__(kBlock) void k_walk_fill(const int *__restrict__ pnode, const int *__restrict__ pimg, long np,
                                                      Layout lay, const unsigned long long *__restrict__ wmask,
                                                      const long *__restrict__ boff, int *__restrict__ cnode,
                                                      int *__restrict__ cimg)
{
    // one lane per PAIR (its 16 children are 16 bits of one count-pass mask): a sixteenth of the lanes of the count pass
    const long p = (long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= np) return;
    constexpr int kPairsPerWave = kWave / kFan, kPairsPerBlock = kBlock / kFan;
    const long cblock = p / kPairsPerBlock;                        // workgroup of the count pass
    const int w = (int)(p % kPairsPerBlock) / kPairsPerWave, sub = (int)(p % kPairsPerWave);
    const unsigned long long mask = wmask[cblock * (kBlock / kWave) + w];
    unsigned bits = (unsigned)(mask >> (kFan * sub)) & ((1u << kFan) - 1u);
    if (!bits) return;
    long pos = wave_offset(wmask, boff, cblock, w) + __popcll(mask & ((1ull << (kFan * sub)) - 1ull));
    const long node = pnode ? pnode[p] : 0;
    const int img = pimg ? pimg[p] : (int)p;
    while (bits) {
        const int j = __ffs(bits) - 1;
        bits &= bits - 1;
        cnode[pos] = (int)child_of(lay, node, j);
        cimg[pos] = img;
        ++pos;
    }
}

// Scratch memory of one weight build.  A build needs a few dozen buffers (two per level of the walk, the records, the sort's
// buffers); through hipMalloc / hipFree every one is a round trip to the driver and hipFree a device synchronisation on top --
// for the 65-transect batch that was more time than the build's kernels take.  So the buffers are cut from a few large
// blocks (32 MiB, doubling): an arena hands out pieces, rewinds when its owner says so (the walk ping-pongs between two
// arenas: the pairs of one level are dead once the next level is written), and frees its blocks when the build ends.
// (Stream-ordered memory pools were tried for this: fine up to 512 transects, but with gigabytes in the pool every later
// hipMalloc took 40 ms -- profiles/r05_weights_scaling.txt.)
struct Arena {
    struct Block {
        char *p;
        size_t size, used;
    };
    std::vector<Block> blocks;
    size_t next_size = 32ull << 20;
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena()
    {
        for (Block &b : blocks) (void)hipFree(b.p);
    }
    void rewind()
    {
        for (Block &b : blocks) b.used = 0;
    }
    hipError_t take(void **out, size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        for (Block &b : blocks)
            if (b.size - b.used >= bytes) {
                *out = b.p + b.used;
                b.used += bytes;
                return hipSuccess;
            }
        Block nb{nullptr, bytes > next_size ? bytes : next_size, 0};
        const hipError_t e = hipMalloc((void **)&nb.p, nb.size);
        if (e != hipSuccess) return e;
        if (next_size < (4ull << 30)) next_size *= 2;
        nb.used = bytes;
        blocks.push_back(nb);
        *out = nb.p;
        return hipSuccess;
    }
    template <typename T> hipError_t take(T **out, size_t count) { return take((void **)out, sizeof(T) * count); }
};

// The arenas of a build, kept between builds (in a process-wide pool, below) while they are small (a viewer makes one
// PolylineIntegral per transect: without this every one of them paid ten hipMalloc / hipFree pairs, more than its kernels).
struct BuildScratch {
    Arena misc, level[2];
    int device = -1;
    size_t capacity() const
    {
        size_t n = 0;
        for (const Arena *a : {&misc, &level[0], &level[1]})
            for (const Arena::Block &b : a->blocks) n += b.size;
        return n;
    }
};
constexpr size_t kScratchKeep = 1024ull << 20;   // the 65-transect batch of the ORCA12-like grid needs 0.9 GiB
constexpr size_t kScratchPool = 4;               // idle scratches kept at most (one per concurrently building host thread)
// The idle scratches of the PROCESS (nf_weights.hip), under a mutex: a build checks one out and hands it back, so a host
// thread that ends leaves nothing behind (round-5 advisor: the per-thread slot leaked up to 1 GiB of HBM per exited thread)
// and nf_release_scratch frees every one of them, whichever thread built with it.
BuildScratch *scratch_checkout(int device);        // an idle scratch of this device, or nullptr
bool scratch_checkin(BuildScratch *sc);            // false: the pool is full, the caller deletes it

struct ScratchLease {   // takes a pooled scratch (or a new one) for one build; gives it back only if told the stream is drained
    BuildScratch *sc = nullptr;
    bool drained = false;
    ScratchLease()
    {
        int dev = -1;
        (void)hipGetDevice(&dev);
        sc = scratch_checkout(dev);
        if (!sc) {
            sc = new BuildScratch();
            sc->device = dev;
        }
    }
    ScratchLease(const ScratchLease &) = delete;
    ScratchLease &operator=(const ScratchLease &) = delete;
    ~ScratchLease()
    {
        if (drained && sc->capacity() <= kScratchKeep) {
            sc->misc.rewind();
            sc->level[0].rewind();
            sc->level[1].rewind();
            if (scratch_checkin(sc)) return;
        }
        delete sc;      // hipFree waits for whatever is still running
    }
};

// ---- the hierarchy of one grid and the breadth-first walk over it, for whoever brings the per-(node, image) test --------------
// (the weight build tests target-segment images, the point location tests points: nf_weights.hip, nf_vinterp.hip)
struct Walker {
    BuildScratch &sc;
    hipStream_t s;
    struct Shape {
        long ty, tx;
    };
    std::vector<Shape> shape;        // shape[l]: the nodes of level l as ty rows of tx (level 0 = the cells)
    std::vector<long> nlev;
    std::vector<Box4 *> boxes;
    int fy = 1, fx = kFan, top = 0;  // a node holds fy x fx nodes of the level below; top: the level of the root
    // per-wavefront masks and per-workgroup counts / offsets of one count-scan-fill step
    unsigned long long *w_mask = nullptr;
    int *w_cnt = nullptr;
    long *w_off = nullptr;
    void *scan_tmp = nullptr;
    long w_cap = 0;
    size_t scan_cap = 0;

    Walker(BuildScratch &scratch, hipStream_t stream) : sc(scratch), s(stream) {}
    // how a node of level l (>= 1) finds its children in level l - 1
    Layout layout(int l) const
    {
        return Layout{(int)shape[(size_t)l - 1].ty, (int)shape[(size_t)l - 1].tx, fy, fx, (int)shape[(size_t)l].tx};
    }
    static long lanes_to_waves(long nlanes) { return ((nlanes + kBlock - 1) / kBlock) * (kBlock / kWave); }

    // The box hierarchy over the cells (level 0 = the cells themselves, a node of level l + 1 = 16 nodes of level l: 4 x 4 blocks
    // when the cells are known to be rows of row_length, 16 consecutive ones otherwise).  keep: the owner's cache, used when it
    // matches, (re)filled when it does not; nullptr: the boxes live in the build's scratch.
    int prepare(const double *xy, long ncell, double period, long row_length, LocatorBoxes *keep)
    {
        const bool tiled = row_length > 1 && ncell % row_length == 0 && ncell / row_length > 1;
        shape.assign(1, tiled ? Shape{ncell / row_length, row_length} : Shape{1, ncell});
        fy = tiled ? 4 : 1;
        fx = tiled ? 4 : kFan;
        while (shape.back().ty * shape.back().tx > 1)
            shape.push_back(Shape{(shape.back().ty + fy - 1) / fy, (shape.back().tx + fx - 1) / fx});
        nlev.clear();
        for (const Shape &q : shape) nlev.push_back(q.ty * q.tx);
        top = (int)nlev.size() - 1;            // the root: one box (top == 0: a one-cell grid)
        boxes.assign(nlev.size(), nullptr);
        const bool cached = keep && keep->xy == xy && keep->ncell == ncell && keep->period == period && keep->count == nlev &&
                            keep->level.size() == nlev.size();
        if (cached) {
            for (size_t l = 0; l < nlev.size(); ++l) boxes[l] = static_cast<Box4 *>(keep->level[l]);
            return NF_OK;
        }
        if (keep) {      // the grid's own copy: plain allocations that outlive this build
            keep->release();
            keep->level.assign(nlev.size(), nullptr);
            keep->count = nlev;
            for (size_t l = 0; l < nlev.size(); ++l) {
                NF_HIP(hipMalloc(&keep->level[l], sizeof(Box4) * (size_t)nlev[l]));
                boxes[l] = static_cast<Box4 *>(keep->level[l]);
            }
        } else {
            for (size_t l = 0; l < nlev.size(); ++l) NF_HIP(sc.misc.take(&boxes[l], (size_t)nlev[l]));
        }
        if (top >= 1)
            hipLaunchKernelGGL(k_boxes_cells, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, xy, ncell, period,
                               boxes[0]);
        for (int l = 1; l <= top; ++l)
            hipLaunchKernelGGL(k_boxes_up, dim3((unsigned)((nlev[(size_t)l] * kFan + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                               (const Box4 *)boxes[(size_t)l - 1], layout(l), nlev[(size_t)l], boxes[(size_t)l]);
        NF_HIP(hipGetLastError());
        if (keep) {
            keep->xy = xy;
            keep->ncell = ncell;
            keep->period = period;
        }
        return NF_OK;
    }

    int reserve_waves(long nw)
    {
        if (nw <= w_cap) return NF_OK;
        nw += nw / 2;                       // the levels grow towards the cells: fewer re-sizes (the old pieces stay in the arena)
        NF_HIP(sc.misc.take(&w_mask, (size_t)nw));
        NF_HIP(sc.misc.take(&w_cnt, (size_t)nw / (kBlock / kWave) + 1));
        NF_HIP(sc.misc.take(&w_off, (size_t)nw / (kBlock / kWave) + 1));
        w_cap = nw;
        return NF_OK;
    }
    // offsets of the workgroups' counts; *total = the number of set bits.  Synchronises the stream.
    int scan_waves(long nwaves, long *total)
    {
        const long nw = nwaves / (kBlock / kWave);     // one count per workgroup
        size_t need = 0;
        NF_HIP(rocprim::exclusive_scan(nullptr, need, w_cnt, w_off, 0l, (size_t)nw, rocprim::plus<long>(), s));
        if (need > scan_cap) {
            NF_HIP(sc.misc.take(&scan_tmp, need + need / 2));
            scan_cap = need + need / 2;
        }
        NF_HIP(rocprim::exclusive_scan(scan_tmp, need, w_cnt, w_off, 0l, (size_t)nw, rocprim::plus<long>(), s));
        long last_off = 0;
        int last_cnt = 0;
        NF_HIP(hipMemcpyAsync(&last_off, w_off + (nw - 1), sizeof(long), hipMemcpyDeviceToHost, s));
        NF_HIP(hipMemcpyAsync(&last_cnt, w_cnt + (nw - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        NF_HIP(hipStreamSynchronize(s));
        *total = last_off + last_cnt;
        return NF_OK;
    }

    // The walk: (group, image) pairs from the root down to (cell, image) candidates, in (image, then walk) order.
    // launch_count(l, pnode, pimg, np, boxes of level l - 1, layout(l), nblocks, w_mask, w_cnt) launches the caller's count kernel:
    // lane t tests child (t % 16) of pair (t / 16) and ends in block_count().  pnode == pimg == nullptr: (root, image p).
    template <class F>
    int walk(long nimg, F launch_count, int **cand_node, int **cand_img, long *ncand)
    {
        int *p_node = nullptr, *p_img = nullptr;
        long np = nimg;
        for (int l = top; l >= 1 && np > 0; --l) {
            const long nlanes = np * kFan;
            NF_REQUIRE(nlanes / kBlock < (1l << 31), NF_ERR_ARG, "locator: too many (cell group, target) pairs; split the set");
            const long nw = lanes_to_waves(nlanes);
            const unsigned nb = (unsigned)(nw / (kBlock / kWave));
            NF_TRY(reserve_waves(nw));
            launch_count(l, (const int *)p_node, (const int *)p_img, np, (const Box4 *)boxes[(size_t)l - 1], layout(l), nb, w_mask,
                         w_cnt);
            long nchild = 0;
            NF_TRY(scan_waves(nw, &nchild));
            // the children go to the other arena: what it held (the parents of this level's parents) is dead.  Kernels that
            // read it have been waited for: scan_waves synchronised the stream after them.
            Arena &dst = sc.level[l & 1];
            dst.rewind();
            int *c_node = nullptr, *c_img = nullptr;
            NF_HIP(dst.take(&c_node, (size_t)nchild));
            NF_HIP(dst.take(&c_img, (size_t)nchild));
            if (nchild > 0)
                hipLaunchKernelGGL(k_walk_fill, dim3((unsigned)((np + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, (const int *)p_node,
                                   (const int *)p_img, np, layout(l), (const unsigned long long *)w_mask, (const long *)w_off, c_node,
                                   c_img);
            NF_HIP(hipGetLastError());
            p_node = c_node;
            p_img = c_img;
            np = nchild;
        }
        if (top == 0 && np > 0) {   // a grid of one cell: every image is a candidate for it
            std::vector<int> zeros((size_t)np, 0), iota((size_t)np);
            for (long k = 0; k < np; ++k) iota[(size_t)k] = (int)k;
            NF_HIP(sc.misc.take(&p_node, (size_t)np));
            NF_HIP(sc.misc.take(&p_img, (size_t)np));
            NF_HIP(hipMemcpyAsync(p_node, zeros.data(), sizeof(int) * (size_t)np, hipMemcpyHostToDevice, s));
            NF_HIP(hipMemcpyAsync(p_img, iota.data(), sizeof(int) * (size_t)np, hipMemcpyHostToDevice, s));
            NF_HIP(hipStreamSynchronize(s));
        }
        *cand_node = p_node;
        *cand_img = p_img;
        *ncand = np;
        return NF_OK;
    }
};

}  // namespace nf
