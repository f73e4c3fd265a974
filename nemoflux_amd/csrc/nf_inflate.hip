// nf_inflate.hip -- file ingest straight to HBM: the deflated (and byte-shuffled) HDF5 chunks of one time step of uo / vo
// are copied to the device as they sit in the NetCDF-4 file and inflated there, one wavefront per chunk, into the staging
// slab the flux kernel reads.
//
// Replaces  nemoflux/field.py:149   nc[name][timeIndex, :, :, :]   -- the lazy NetCDF read, i.e. HDF5's filter pipeline
//           (deflate, then un-shuffle) that netCDF4/xarray run on the host for every time step.
//
// The decoder itself is nf_inflate_core.h (RFC 1950/1951, Adler-32 checked).  One workgroup = one wavefront = one stream;
// its 39 KiB state (32 KiB window, input ring, lookup tables) lives in LDS, so four streams run per CU and a thousand
// at once on the chip -- the format is serial inside a stream, the parallelism is across the chunks (75 levels x 2 fields
// per time step in XIOS output).  HDF5's shuffle filter stored the bytes of every element de-interleaved (all first bytes,
// then all second bytes, ...): k_place gathers them back, one element per lane, coalesced on both sides, and puts the
// chunk where it belongs in the (nz, ny, nx) slab (chunks may tile y and x, edge chunks hang over).
#include <type_traits>
#include <algorithm>
#include <vector>

#include "nf_common.h"
#include "nf_inflate_core.h"

#define NF_TRY_RC(call)                \
    do {                              \
        int rc_ = (call);             \
        if (rc_ != NF_OK) return rc_; \
    } while (0)

namespace nf {

struct InflateJob {
    unsigned long long in_off;    // first byte of the zlib stream in the compressed buffer
    unsigned in_len;
    unsigned z0, y0, x0;          // origin of the chunk in the slab (elements)
};

// inflate chunk i into slot i of tmp (chunk_bytes each)
__global__ __launch_bounds__(64) void k_inflate(const uint8_t *__restrict__ comp, unsigned long long comp_bytes,
                                                const InflateJob *__restrict__ jobs, int njobs, uint8_t *__restrict__ tmp,
                                                unsigned chunk_bytes, int *__restrict__ status)
{
    __shared__ NfiCtx ctx;
    const int i = blockIdx.x;
    if (i >= njobs) return;
    if ((uint32_t)(uintptr_t)&ctx != 0u) {      // the decoder's hand-written loop addresses ctx from LDS offset 0
        if (threadIdx.x == 0) status[i] = NFI_ERR_LAYOUT;
        return;
    }
    const InflateJob job = jobs[i];
    const unsigned long long first = job.in_off & ~3ull;                 // the word that holds the stream's first byte
    unsigned long long readable = comp_bytes - first;                    // comp_bytes is a multiple of 4 (padded by the host)
    const unsigned long long want = (job.in_off - first) + job.in_len + 16ull;
    if (readable > want) readable = want & ~3ull;
    const int rc = nfi_inflate_stream(ctx, comp + job.in_off, job.in_len, (uint32_t)readable,
                                      tmp + (unsigned long long)i * chunk_bytes, chunk_bytes);
    if (threadIdx.x == 0) status[i] = rc;
}

// Inverse of HDF5's shuffle filter + placement of the chunk in the slab; blockIdx.y walks the chunks.  A shuffled chunk holds
// the ES byte planes of its n elements one after the other; an element of the chunk at (a, b, c) of its (cz, cy, cx) box
// goes to ((z0+a)*ny + y0+b)*nx + x0+c of the (nz, ny, nx) slab; the parts of an edge chunk that hang over the slab are
// dropped.  A lane owns FOUR consecutive elements of a row: one 4-byte load per byte plane (a wave reads 256 contiguous
// bytes of every plane), a byte transpose in registers, one 16- or 32-byte store.  Chunks whose rows are not a multiple of
// four elements long take the one-element-per-lane form.
struct SlabGeom {
    unsigned cz, cy, cx, nz, ny, nx;
};
template <int ES, bool SHUFFLED>
__global__ __launch_bounds__(kBlock) void k_place(const uint8_t *__restrict__ tmp, unsigned chunk_bytes,
                                                  const InflateJob *__restrict__ jobs, int njobs, SlabGeom g,
                                                  uint8_t *__restrict__ dst)
{
    const unsigned long long n = (unsigned long long)g.cz * g.cy * g.cx;
    for (int i = blockIdx.y; i < njobs; i += gridDim.y) {
        const InflateJob job = jobs[i];
        const uint8_t *s = tmp + (unsigned long long)i * chunk_bytes;
        for (unsigned long long e = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; e < n;
             e += (unsigned long long)gridDim.x * kBlock) {
            const unsigned c = (unsigned)(e % g.cx);
            const unsigned long long r = e / g.cx;
            const unsigned b = (unsigned)(r % g.cy), a = (unsigned)(r / g.cy);
            const unsigned z = job.z0 + a, y = job.y0 + b, x = job.x0 + c;
            if (z >= g.nz || y >= g.ny || x >= g.nx) continue;
            const unsigned long long o = ((unsigned long long)z * g.ny + y) * g.nx + x;
            if (ES == 1) {
                dst[o] = s[e];
            } else if (ES == 4) {
                uint32_t v;
                if (SHUFFLED) v = (uint32_t)s[e] | ((uint32_t)s[n + e] << 8) | ((uint32_t)s[2 * n + e] << 16) | ((uint32_t)s[3 * n + e] << 24);
                else v = reinterpret_cast<const uint32_t *>(s)[e];
                reinterpret_cast<uint32_t *>(dst)[o] = v;
            } else {
                uint64_t v = 0;
                if (SHUFFLED) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v |= (uint64_t)s[(unsigned long long)k * n + e] << (8 * k);
                } else {
                    v = reinterpret_cast<const uint64_t *>(s)[e];
                }
                reinterpret_cast<uint64_t *>(dst)[o] = v;
            }
        }
    }
}

// four elements per lane; needs cx % 4 == 0 (so n % 4 == 0 and a lane's elements share a row).  PLANES: every chunk holds
// whole (y, x) planes (cy == ny, cx == nx -- one chunk per level, what XIOS writes): element e of the chunk is element
// z0*ny*nx + e of the slab, no index arithmetic.  A chunk holds fewer than 2^31 bytes: 32-bit indices throughout.
template <int ES, bool PLANES>
__global__ __launch_bounds__(kBlock) void k_place4(const uint8_t *__restrict__ tmp, unsigned chunk_bytes,
                                                   const InflateJob *__restrict__ jobs, int njobs, SlabGeom g,
                                                   uint8_t *__restrict__ dst)
{
    using elem_t = typename std::conditional<ES == 4, uint32_t, uint64_t>::type;
    const unsigned n = g.cz * g.cy * g.cx, nq = n / 4;
    const unsigned plane = g.ny * g.nx;
    for (int i = blockIdx.y; i < njobs; i += gridDim.y) {
        const InflateJob job = jobs[i];
        const uint32_t *s = reinterpret_cast<const uint32_t *>(tmp + (unsigned long long)i * chunk_bytes);
        for (unsigned q = blockIdx.x * kBlock + threadIdx.x; q < nq; q += gridDim.x * kBlock) {
            const unsigned e = 4 * q;
            unsigned x, y, z;
            if (PLANES) {
                z = job.z0 + e / plane;           // only the bound matters: the level an over-hanging chunk must stop at
                y = 0;
                x = 0;
                if (z >= g.nz) continue;
            } else {
                const unsigned c = e % g.cx, r = e / g.cx;
                const unsigned b = r % g.cy, a = r / g.cy;
                z = job.z0 + a;
                y = job.y0 + b;
                x = job.x0 + c;
                if (z >= g.nz || y >= g.ny || x >= g.nx) continue;
            }
            uint32_t w[ES];
#pragma unroll
            for (int p = 0; p < ES; ++p) w[p] = __builtin_nontemporal_load(s + (p * n + e) / 4);   // bytes p of elements e .. e+3
            elem_t out[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                elem_t v = 0;
#pragma unroll
                for (int p = 0; p < ES; ++p) v |= (elem_t)((w[p] >> (8 * k)) & 255u) << (8 * p);
                out[k] = v;
            }
            const unsigned long long o = PLANES ? (unsigned long long)job.z0 * plane + e
                                                : ((unsigned long long)z * g.ny + y) * g.nx + x;
            elem_t *d = reinterpret_cast<elem_t *>(dst) + o;
            if ((PLANES || x + 3 < g.nx) && (o & 3ull) == 0) {   // whole and 16-byte aligned (the usual case)
                typedef elem_t vec4 __attribute__((ext_vector_type(4)));
                const vec4 v = {out[0], out[1], out[2], out[3]};
                *reinterpret_cast<vec4 *>(d) = v;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k < g.nx) d[k] = out[k];
            }
        }
    }
}

// float32, whole (y, x) planes per chunk, plane size a multiple of 16 elements -- one deflated chunk per level of a NEMO
// file: a lane owns SIXTEEN consecutive elements: one 16-byte load per byte plane (a wave reads 1 KiB contiguous of each),
// a byte transpose in registers, four 16-byte stores (a wave writes 4 KiB contiguous).
__global__ __launch_bounds__(kBlock) void k_place16(const uint8_t *__restrict__ tmp, unsigned chunk_bytes,
                                                    const InflateJob *__restrict__ jobs, int njobs, SlabGeom g,
                                                    uint8_t *__restrict__ dst)
{
    typedef uint32_t uvec4 __attribute__((ext_vector_type(4)));
    const unsigned n = g.cz * g.cy * g.cx, nq = n / 16, plane = g.ny * g.nx;
    for (int i = blockIdx.y; i < njobs; i += gridDim.y) {
        const InflateJob job = jobs[i];
        const uint8_t *s = tmp + (unsigned long long)i * chunk_bytes;
        for (unsigned q = blockIdx.x * kBlock + threadIdx.x; q < nq; q += gridDim.x * kBlock) {
            const unsigned e = 16 * q;
            if (job.z0 + e / plane >= g.nz) continue;            // the levels of an over-hanging chunk that lie beyond the slab
            uvec4 w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) w[p] = __builtin_nontemporal_load(reinterpret_cast<const uvec4 *>(s + (unsigned long long)p * n + e));
            uvec4 *d = reinterpret_cast<uvec4 *>(reinterpret_cast<uint32_t *>(dst) + (unsigned long long)job.z0 * plane + e);
#pragma unroll
            for (int c = 0; c < 4; ++c) {                        // word c of every plane holds elements 4c .. 4c+3
                uvec4 out;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    out[k] = ((w[0][c] >> (8 * k)) & 255u) | (((w[1][c] >> (8 * k)) & 255u) << 8) |
                             (((w[2][c] >> (8 * k)) & 255u) << 16) | (((w[3][c] >> (8 * k)) & 255u) << 24);
                d[c] = out;
            }
        }
    }
}

struct Inflater {
    uint8_t *d_comp = nullptr, *d_tmp = nullptr;
    InflateJob *d_jobs = nullptr;
    int *d_status = nullptr;
    size_t comp_cap = 0, tmp_cap = 0, jobs_cap = 0;
    // early upload (nf_inflater_upload, typically from a staging thread while the GPU decodes another group)
    hipStream_t copy_stream = nullptr;
    size_t uploaded = 0;          // bytes of compressed data sitting in d_comp, 0 = none
    int device = -1;
    // nf_inflater_share_scratch: the decode scratch (d_tmp = the whole decoded group, d_jobs, d_status) of ANOTHER inflater is
    // used instead of an own one.  Only the compressed buffer needs one copy per staging slot (it is filled by the staging
    // thread while the other slot decodes); a run is synchronous on its caller's thread, so two inflaters whose runs never
    // overlap in time can share the rest (a file-backed Field then holds one decoded-group scratch, not two).
    Inflater *scratch_owner = nullptr;
    std::vector<Inflater *> borrowers;   // the inflaters that use THIS one's scratch: detached when this one is deleted
    void release()
    {
        for (void *p : {(void *)d_comp, (void *)d_tmp, (void *)d_jobs, (void *)d_status})
            if (p) (void)hipFree(p);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        copy_stream = nullptr;
        d_comp = d_tmp = nullptr;
        d_jobs = nullptr;
        d_status = nullptr;
        comp_cap = tmp_cap = jobs_cap = uploaded = 0;
    }
};

static size_t comp_padded(size_t comp_bytes) { return ((comp_bytes + 3) & ~(size_t)3) + 64; }   // whole words + slack the ring may read

static int comp_reserve(Inflater *h, size_t comp_pad)
{
    if (h->comp_cap < comp_pad) {
        if (h->d_comp) (void)hipFree(h->d_comp);
        h->d_comp = nullptr;
        h->comp_cap = 0;
        h->uploaded = 0;
        NF_HIP(hipMalloc((void **)&h->d_comp, comp_pad));
        h->comp_cap = comp_pad;
    }
    return NF_OK;
}

// compressed bytes -> HBM, on the inflater's own stream, complete at return: the call a staging thread makes while the GPU
// is busy with the previous group (HIP calls are thread-safe; the thread adopts the inflater's device)
int inflater_upload(Inflater *h, const void *comp_host, size_t comp_bytes)
{
    NF_REQUIRE(h && comp_host && comp_bytes > 0, NF_ERR_ARG, "inflate upload: null argument");
    if (h->device >= 0) NF_HIP(hipSetDevice(h->device));
    if (!h->copy_stream) NF_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    const size_t comp_pad = comp_padded(comp_bytes);
    h->uploaded = 0;
    NF_TRY_RC(comp_reserve(h, comp_pad));
    const size_t tail = comp_bytes & ~(size_t)3;
    NF_HIP(hipMemsetAsync(h->d_comp + tail, 0, comp_pad - tail, h->copy_stream));
    NF_HIP(hipMemcpyAsync(h->d_comp, comp_host, comp_bytes, hipMemcpyHostToDevice, h->copy_stream));
    NF_HIP(hipStreamSynchronize(h->copy_stream));
    h->uploaded = comp_bytes;
    return NF_OK;
}

static const char *inflate_error_name(int rc)
{
    switch (rc) {
        case NFI_ERR_HEADER: return "not a zlib stream";
        case NFI_ERR_BLOCK: return "bad block header";
        case NFI_ERR_CODES: return "invalid Huffman code set";
        case NFI_ERR_SYMBOL: return "invalid code in the data";
        case NFI_ERR_DISTANCE: return "match distance before the start of the chunk";
        case NFI_ERR_OUTPUT: return "decoded length differs from the chunk size";
        case NFI_ERR_INPUT: return "compressed stream is truncated";
        case NFI_ERR_CHECKSUM: return "Adler-32 mismatch";
        case NFI_ERR_LAYOUT: return "decoder state is not at LDS offset 0 (build problem)";
        default: return "unknown error";
    }
}

// The same without a staging copy: n byte ranges of (pageable) host memory -- the compressed chunks as they sit in the mapped
// file -- go to their places in d_comp one hipMemcpyAsync each; the runtime pipelines its own bounce buffers with the DMA, so
// the host-side gather and the transfer overlap instead of running one after the other.  Complete at return.
int inflater_upload_ranges(Inflater *h, const unsigned long long *src_addr, const long long *dst_off, const long long *len,
                           long long n, size_t comp_bytes)
{
    NF_REQUIRE(h && src_addr && dst_off && len && n > 0 && comp_bytes > 0, NF_ERR_ARG, "inflate upload: null argument");
    if (h->device >= 0) NF_HIP(hipSetDevice(h->device));
    if (!h->copy_stream) NF_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    const size_t comp_pad = comp_padded(comp_bytes);
    h->uploaded = 0;
    NF_TRY_RC(comp_reserve(h, comp_pad));
    for (long long i = 0; i < n; ++i)
        NF_REQUIRE(len[i] >= 0 && dst_off[i] >= 0 && (size_t)(dst_off[i] + len[i]) <= comp_bytes, NF_ERR_ARG,
                   "inflate upload: a range lies outside the compressed buffer");
    NF_HIP(hipMemsetAsync(h->d_comp, 0, comp_pad, h->copy_stream));      // gaps between the ranges and the padding read as zeros
    for (long long i = 0; i < n; ++i)
        if (len[i])
            NF_HIP(hipMemcpyAsync(h->d_comp + dst_off[i], (const void *)(uintptr_t)src_addr[i], (size_t)len[i],
                                  hipMemcpyHostToDevice, h->copy_stream));
    NF_HIP(hipStreamSynchronize(h->copy_stream));
    h->uploaded = comp_bytes;
    return NF_OK;
}

int inflater_run(Inflater *h, const void *comp_host, size_t comp_bytes, const long long *in_off, const long long *in_len,
                 int n, long long chunk_bytes, int elem_size, int shuffled, const long long *chunk_dims,
                 const long long *slab_dims, const long long *origin, void *out_dev, hipStream_t s, int *status_host)
{
    NF_REQUIRE(h && chunk_dims && slab_dims && (n == 0 || (in_off && in_len && origin && out_dev)), NF_ERR_ARG,
               "inflate: null argument");
    NF_REQUIRE(n == 0 || comp_host || (h->uploaded == comp_bytes && comp_bytes > 0), NF_ERR_STATE,
               "inflate: no compressed buffer given and none of that size uploaded (nf_inflater_upload)");
    NF_REQUIRE(elem_size == 1 || elem_size == 4 || elem_size == 8, NF_ERR_ARG, "inflate: element size must be 1, 4 or 8");
    NF_REQUIRE(!(shuffled && elem_size == 1), NF_ERR_ARG, "inflate: single bytes cannot be shuffled");
    for (int k = 0; k < 3; ++k)
        NF_REQUIRE(chunk_dims[k] > 0 && slab_dims[k] > 0 && chunk_dims[k] < (1ll << 31) && slab_dims[k] < (1ll << 31), NF_ERR_ARG,
                   "inflate: bad chunk / slab dimensions");
    NF_REQUIRE(chunk_bytes == chunk_dims[0] * chunk_dims[1] * chunk_dims[2] * elem_size && chunk_bytes < (1ll << 31), NF_ERR_ARG,
               "inflate: chunk_bytes does not match the chunk dimensions (chunks of up to 2 GiB)");
    if (n == 0) return NF_OK;
    {   // the decoder's phases are ordered by the issue order of ONE wavefront: a 64-lane workgroup must be exactly that
        int dev = 0, wave = 0;
        NF_HIP(hipGetDevice(&dev));
        NF_HIP(hipDeviceGetAttribute(&wave, hipDeviceAttributeWarpSize, dev));
        NF_REQUIRE(wave == 64, NF_ERR_NO_DEVICE, "inflate: the device decoder needs 64-lane wavefronts (CDNA)");
    }
    std::vector<InflateJob> jobs((size_t)n);
    for (int i = 0; i < n; ++i) {
        NF_REQUIRE(in_off[i] >= 0 && in_len[i] >= 0 && (size_t)(in_off[i] + in_len[i]) <= comp_bytes && in_len[i] < (1ll << 31),
                   NF_ERR_ARG, "inflate: a compressed chunk lies outside the buffer");
        for (int k = 0; k < 3; ++k)
            NF_REQUIRE(origin[3 * i + k] >= 0 && origin[3 * i + k] < slab_dims[k], NF_ERR_ARG,
                       "inflate: a chunk starts outside the slab");
        jobs[i] = InflateJob{(unsigned long long)in_off[i], (unsigned)in_len[i], (unsigned)origin[3 * i], (unsigned)origin[3 * i + 1],
                             (unsigned)origin[3 * i + 2]};
    }
    const size_t comp_pad = comp_padded(comp_bytes);
    const size_t tmp_bytes = (size_t)chunk_bytes * (size_t)n;
    if (comp_host) NF_TRY_RC(comp_reserve(h, comp_pad));
    Inflater *so = h->scratch_owner ? h->scratch_owner : h;    // whose decode scratch this run uses
    if (so->tmp_cap < tmp_bytes) {
        if (so->d_tmp) (void)hipFree(so->d_tmp);
        so->d_tmp = nullptr;
        so->tmp_cap = 0;
        NF_HIP(hipMalloc((void **)&so->d_tmp, tmp_bytes));
        so->tmp_cap = tmp_bytes;
    }
    if (so->jobs_cap < (size_t)n) {
        if (so->d_jobs) (void)hipFree(so->d_jobs);
        if (so->d_status) (void)hipFree(so->d_status);
        so->d_jobs = nullptr;
        so->d_status = nullptr;
        so->jobs_cap = 0;
        NF_HIP(hipMalloc((void **)&so->d_jobs, sizeof(InflateJob) * (size_t)n));
        NF_HIP(hipMalloc((void **)&so->d_status, sizeof(int) * (size_t)n));
        so->jobs_cap = (size_t)n;
    }
    // From here on work is queued on `s` that reads `jobs` (pageable host memory) and writes the scratch -- which another
    // inflater may share and use from ITS stream next: whatever makes this function return early, the stream is drained first
    struct DrainOnExit {
        hipStream_t s;
        ~DrainOnExit() { (void)hipStreamSynchronize(s); }
    } drain{s};
    if (comp_host) {
        const size_t tail = comp_bytes & ~(size_t)3;                    // zero the last partial word and the padding behind the data
        h->uploaded = 0;
        NF_HIP(hipMemsetAsync(h->d_comp + tail, 0, comp_pad - tail, s));
        NF_HIP(hipMemcpyAsync(h->d_comp, comp_host, comp_bytes, hipMemcpyHostToDevice, s));
    }
    NF_HIP(hipMemcpyAsync(so->d_jobs, jobs.data(), sizeof(InflateJob) * (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)n), dim3(64), 0, s, h->d_comp, (unsigned long long)comp_pad,
                       so->d_jobs, n, so->d_tmp, (unsigned)chunk_bytes, so->d_status);
    const SlabGeom g{(unsigned)chunk_dims[0], (unsigned)chunk_dims[1], (unsigned)chunk_dims[2], (unsigned)slab_dims[0],
                     (unsigned)slab_dims[1], (unsigned)slab_dims[2]};
    const bool four = shuffled && chunk_dims[2] % 4 == 0;             // four elements per lane (k_place4)
    // whole (y, x) planes per chunk whose plane size is a multiple of four elements: no index arithmetic at all
    const bool planes = chunk_dims[1] == slab_dims[1] && chunk_dims[2] == slab_dims[2] && (slab_dims[1] * slab_dims[2]) % 4 == 0;
    const bool sixteen = elem_size == 4 && four && planes && (slab_dims[1] * slab_dims[2]) % 16 == 0;
    const long long nelem = chunk_bytes / elem_size / (sixteen ? 16 : four ? 4 : 1);
    unsigned gx = (unsigned)std::min<long long>(4096, (nelem + kBlock - 1) / kBlock);
    if (gx == 0) gx = 1;
    const dim3 grid(gx, (unsigned)std::min(n, 65535)), block(kBlock);  // gridDim.y is capped: the kernels walk the chunks
    uint8_t *dst = (uint8_t *)out_dev;
    if (elem_size == 1) hipLaunchKernelGGL((k_place<1, false>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (sixteen) hipLaunchKernelGGL(k_place16, grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (elem_size == 4 && four && planes) hipLaunchKernelGGL((k_place4<4, true>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (elem_size == 8 && four && planes) hipLaunchKernelGGL((k_place4<8, true>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (elem_size == 4 && four) hipLaunchKernelGGL((k_place4<4, false>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (elem_size == 8 && four) hipLaunchKernelGGL((k_place4<8, false>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (elem_size == 4 && shuffled) hipLaunchKernelGGL((k_place<4, true>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (elem_size == 4) hipLaunchKernelGGL((k_place<4, false>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else if (shuffled) hipLaunchKernelGGL((k_place<8, true>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    else hipLaunchKernelGGL((k_place<8, false>), grid, block, 0, s, so->d_tmp, (unsigned)chunk_bytes, so->d_jobs, n, g, dst);
    NF_HIP(hipGetLastError());
    std::vector<int> status((size_t)n, 0);
    NF_HIP(hipMemcpyAsync(status.data(), so->d_status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    NF_HIP(hipStreamSynchronize(s));    // jobs / status vectors are pageable host memory: the copies above are done now
    int bad = -1;
    for (int i = 0; i < n; ++i) {
        if (status_host) status_host[i] = status[i];
        if (status[i] != 0 && bad < 0) bad = i;
    }
    if (bad >= 0) {
        char buf[200];
        snprintf(buf, sizeof buf, "inflate: chunk %d of %d: %s (code %d)", bad, n, inflate_error_name(status[bad]), status[bad]);
        set_error(buf);
        return NF_ERR_ARG;
    }
    return NF_OK;
}

}  // namespace nf

// ------------------------------------------------------------------------------------------------------- C ABI
using namespace nf;
extern "C" {

int nf_inflater_share_scratch(nf_inflater **self, nf_inflater **owner)
{
    if (!self || !*self || !owner || !*owner || *self == *owner) {
        set_error("nf_inflater_share_scratch: two different inflaters are needed");
        return NF_ERR_ARG;
    }
    Inflater *h = reinterpret_cast<Inflater *>(*self), *o = reinterpret_cast<Inflater *>(*owner);
    if (o->scratch_owner) {
        set_error("nf_inflater_share_scratch: the owner borrows its scratch itself");
        return NF_ERR_ARG;
    }
    if (h->device != o->device) {     // the scratch is device memory: a run on another GPU could not even address it
        set_error("nf_inflater_share_scratch: the two inflaters were created on different devices");
        return NF_ERR_ARG;
    }
    if (!h->borrowers.empty()) {
        set_error("nf_inflater_share_scratch: this inflater lends its scratch to others");
        return NF_ERR_ARG;
    }
    try {
        if (h->scratch_owner != o) o->borrowers.push_back(h);
    } catch (...) {
        set_error("out of host memory");
        return NF_ERR_HOST;
    }
    if (h->scratch_owner && h->scratch_owner != o) {     // it borrowed from someone else before
        auto &b = h->scratch_owner->borrowers;
        b.erase(std::remove(b.begin(), b.end(), h), b.end());
    }
    for (void *p : {(void *)h->d_tmp, (void *)h->d_jobs, (void *)h->d_status})   // its own scratch is not needed any more
        if (p) (void)hipFree(p);
    h->d_tmp = nullptr;
    h->d_jobs = nullptr;
    h->d_status = nullptr;
    h->tmp_cap = h->jobs_cap = 0;
    h->scratch_owner = o;
    return NF_OK;
}

int nf_inflater_new(nf_inflater **self)
{
    if (!self) {
        set_error("nf_inflater_new: null argument");
        return NF_ERR_ARG;
    }
    Inflater *h = new (std::nothrow) Inflater();
    *self = reinterpret_cast<nf_inflater *>(h);
    if (!h) {
        set_error("out of host memory");
        return NF_ERR_HOST;
    }
    if (hipGetDevice(&h->device) != hipSuccess) h->device = -1;      // no GPU: every run fails loudly later
    return NF_OK;
}

int nf_inflater_upload_ranges(nf_inflater **self, const unsigned long long *src_addr, const long long *dst_off,
                              const long long *len, long long n, size_t comp_bytes)
{
    if (!self || !*self) {
        set_error("nf_inflater_upload_ranges: null handle");
        return NF_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no usable AMD GPU (hipGetDeviceCount); nemoflux_amd has no CPU fallback");
        return NF_ERR_NO_DEVICE;
    }
    try {
        return inflater_upload_ranges(reinterpret_cast<Inflater *>(*self), src_addr, dst_off, len, n, comp_bytes);
    } catch (...) {
        set_error("nf_inflater_upload_ranges: out of host memory");
        return NF_ERR_HOST;
    }
}

int nf_inflater_upload(nf_inflater **self, const void *comp_host, size_t comp_bytes)
{
    if (!self || !*self) {
        set_error("nf_inflater_upload: null handle");
        return NF_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no usable AMD GPU (hipGetDeviceCount); nemoflux_amd has no CPU fallback");
        return NF_ERR_NO_DEVICE;
    }
    try {
        return inflater_upload(reinterpret_cast<Inflater *>(*self), comp_host, comp_bytes);
    } catch (...) {
        set_error("nf_inflater_upload: out of host memory");
        return NF_ERR_HOST;
    }
}

/* wavefronts (= chunks) the device decodes at once: resident k_inflate workgroups per CU (LDS-limited) x CUs */
int nf_inflater_capacity(int *streams)
{
    if (!streams) {
        set_error("nf_inflater_capacity: null argument");
        return NF_ERR_ARG;
    }
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_inflate, 64, 0) != hipSuccess) {
        set_error("no usable AMD GPU (hipGetDevice); nemoflux_amd has no CPU fallback");
        return NF_ERR_NO_DEVICE;
    }
    *streams = (per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;
    return NF_OK;
}

int nf_inflater_del(nf_inflater **self)
{
    if (self && *self) {
        Inflater *h = reinterpret_cast<Inflater *>(*self);
        // nobody keeps a pointer to a deleted inflater: its borrowers go back to a scratch of their own (allocated at their
        // next run), and it leaves the list of the one it borrowed from.  Kernels of a borrower may still be reading the
        // scratch (its runs are synchronous, but an error return can leave work in flight): wait before freeing it
        if (!h->borrowers.empty()) (void)hipDeviceSynchronize();
        for (Inflater *b : h->borrowers) b->scratch_owner = nullptr;
        if (h->scratch_owner) {
            auto &b = h->scratch_owner->borrowers;
            b.erase(std::remove(b.begin(), b.end(), h), b.end());
        }
        h->release();
        delete h;
        *self = nullptr;
    }
    return NF_OK;
}

int nf_inflater_run(nf_inflater **self, const void *comp_host, size_t comp_bytes, const long long *in_off,
                    const long long *in_len, int nchunks, long long chunk_bytes, int elem_size, int shuffled,
                    const long long *chunk_dims, const long long *slab_dims, const long long *origin, void *out_dev,
                    void *hip_stream, int *status_host)
{
    if (!self || !*self) {
        set_error("nf_inflater_run: null handle");
        return NF_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no usable AMD GPU (hipGetDeviceCount); nemoflux_amd has no CPU fallback");
        return NF_ERR_NO_DEVICE;
    }
    try {
        return inflater_run(reinterpret_cast<Inflater *>(*self), comp_host, comp_bytes, in_off, in_len, nchunks, chunk_bytes,
                            elem_size, shuffled, chunk_dims, slab_dims, origin, out_dev, (hipStream_t)hip_stream, status_host);
    } catch (...) {
        set_error("nf_inflater_run: out of host memory");
        return NF_ERR_HOST;
    }
}

}  // extern "C"
