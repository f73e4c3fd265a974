// nf_integral.hip -- K3: gather of the weighted edge fluxes and their reduction onto target segments.
//
// Replaces  mint.PolylineIntegral.getIntegral(data, mint.CELL_BY_CELL_DATA) as driven by
//           nemoflux/field.py:102 and nemoflux/fluxplot.py:56, for ALL transects of a Field at once.
//
// Two forms of stage A.  For the engine's own resident planes the weights come folded onto the unique edges
// (k_gather_edges, below: one gather per entry); for caller-supplied (ncell,4) data -- mint's getIntegral -- the records
// are used as they are:
// Records (cell, 4 edge weights, global segment id) are sorted by segment (K2).  Stage A: one lane per
// record gathers the cell's 4 edge values (either the reference's (ncell,4) AoS: one 32-B read, or the engine's
// resident [4][ncell] planes), forms the weighted sum and runs a WAVEFRONT SEGMENTED SCAN keyed by the
// segment id (6 shuffle steps); the last lane of every run inside the wave stores the run's sum.  Stage B: one
// wavefront per target segment stitches the run sums of the waves the segment spans.  Stage C: one wavefront
// per transect adds its segments.  No atomics: the summation tree is fixed, so results are bitwise
// reproducible.
//
// Output row: [ per-segment sums (nseg) | per-transect sums (ntransect) ].
#include "nf_common.h"

namespace nf {

__global__ __launch_bounds__(kBlock) void k_gather_segscan(const int *__restrict__ cell,
                                                           const double *__restrict__ w4,
                                                           const int *__restrict__ seg, long n,
                                                           const double *__restrict__ data, long ncell,
                                                           int planes, unsigned nx, double *__restrict__ runsum,
                                                           long data_stride)
{
    data += (long)blockIdx.y * data_stride;  // blockIdx.y = time step of a multi-step launch
    runsum += (long)blockIdx.y * n;
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    double val = 0.0;
    int key = -1;
    if (k < n) {
        // the record stream is read once per launch: non-temporal, so that it does not push the plane sectors its
        // neighbours gather -- nor the arc lengths the next flux kernel re-reads -- out of the caches (-0.7 % per pass)
        typedef double dvec2 __attribute__((ext_vector_type(2)));
        const long c = __builtin_nontemporal_load(cell + k);
        const dvec2 *pw = reinterpret_cast<const dvec2 *>(w4 + 4 * k);
        const dvec2 wa = __builtin_nontemporal_load(pw), wb = __builtin_nontemporal_load(pw + 1);
        double d0, d1, d2, d3;
        if (planes == 2) {
            // the engine's own planes: the south and west slots are copies of the neighbours' north and east values
            // (field.py:219-223), so only the two signed planes are touched: eU[c-1], eU[c] share a sector and
            // eV[c-nx] is the eV[c] of the record one row below -- about half the sectors of four separate planes
            const unsigned j = (unsigned)(c / nx), i = (unsigned)(c - (long)j * nx);
            const double *eU = data + ncell, *eV = data + 2 * ncell;
            d1 = eU[c];
            d2 = eV[c];
            d0 = j > 0 ? eV[c - nx] : 0.0;               // row 0's south slot is never written (field.py:219)
            d3 = eU[i > 0 ? c - 1 : c - 1 + nx];         // column 0: periodic copy of column nx-1 (field.py:223)
        } else if (planes) {
            d0 = data[c];
            d1 = data[ncell + c];
            d2 = data[2 * ncell + c];
            d3 = data[3 * ncell + c];
        } else {
            const double2 *pd = reinterpret_cast<const double2 *>(data + 4 * c);
            const double2 da = pd[0], db = pd[1];
            d0 = da.x; d1 = da.y; d2 = db.x; d3 = db.y;
        }
        val = ((wa.x * d0 + wa.y * d1) + wb.x * d2) + wb.y * d3;
        key = __builtin_nontemporal_load(seg + k);
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const double pv = __shfl_up(val, o, kWave);
        const int pk = __shfl_up(key, o, kWave);
        if (lane >= o && pk == key) val += pv;
    }
    const int nk = __shfl_down(key, 1, kWave);
    if (k < n && (lane == kWave - 1 || k == n - 1 || nk != key)) runsum[k] = val;
}

// Unique-edge form of stage A for the engine's own planes (WeightSet::EdgeEntry): one lane per (segment, plane element)
// entry -- ONE 16-byte record load and ONE 8-byte gather from [eU | eV] -- then the same wavefront segmented scan.
// Entries are sorted by (segment, element), so the gathers of neighbouring lanes walk the planes in ascending order.
__global__ __launch_bounds__(kBlock) void k_gather_edges(const WeightSet::EdgeEntry *__restrict__ ent, long n,
                                                         const double *__restrict__ data, long ncell,
                                                         double *__restrict__ runsum, long data_stride)
{
    data += (long)blockIdx.y * data_stride + ncell;   // planes 1 (eU) and 2 (eV) are contiguous: one array of 2*ncell
    runsum += (long)blockIdx.y * n;
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    double val = 0.0;
    int key = -1;
    if (k < n) {
        typedef int ivec4 __attribute__((ext_vector_type(4)));
        const ivec4 r = __builtin_nontemporal_load(reinterpret_cast<const ivec4 *>(ent + k));   // read-once stream
        double w;
        const int wbits[2] = {r.z, r.w};
        __builtin_memcpy(&w, wbits, 8);
        val = w * data[(unsigned)r.x];
        key = r.y;
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const double pv = __shfl_up(val, o, kWave);
        const int pk = __shfl_up(key, o, kWave);
        if (lane >= o && pk == key) val += pv;
    }
    const int nk = __shfl_down(key, 1, kWave);
    if (k < n && (lane == kWave - 1 || k == n - 1 || nk != key)) runsum[k] = val;
}

// one wavefront per target segment: stitch the per-wave run sums of the segment (one per 64-record wave it
// spans) with a lane-strided sum and a butterfly
__global__ __launch_bounds__(kBlock) void k_finalize_seg(const double *__restrict__ runsum,
                                                         const int *__restrict__ seg_start, int nseg,
                                                         double *__restrict__ row, long nrec, long row_stride)
{
    runsum += (long)blockIdx.y * nrec;
    row += (long)blockIdx.y * row_stride;
    const int s = (blockIdx.x * kBlock + threadIdx.x) / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (s >= nseg) return;
    const long lo = seg_start[s], hi = seg_start[s + 1];
    double acc = 0.0;
    if (hi > lo) {
        const long w0 = lo / kWave, w1 = (hi - 1) / kWave;  // waves of stage A touched by this segment
        for (long w = w0 + lane; w <= w1; w += kWave) {
            long e = (w + 1) * kWave;
            if (e > hi) e = hi;
            acc += runsum[e - 1];
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
    if (lane == 0) row[s] = acc;
}

// one wavefront per transect: sum of its segments
__global__ __launch_bounds__(kBlock) void k_finalize_tr(const int *__restrict__ tr_off, int ntransect, int nseg,
                                                        double *__restrict__ row, long row_stride)
{
    row += (long)blockIdx.y * row_stride;
    const int p = (blockIdx.x * kBlock + threadIdx.x) / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (p >= ntransect) return;
    double part = 0.0;
    for (int s = tr_off[p] + lane; s < tr_off[p + 1]; s += kWave) part += row[s];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, kWave);
    if (lane == 0) row[nseg + p] = part;
}

// "edge_weights" tuning knob.  Default 0: measured in-process on the 65-transect bench batch (tools/ab_pass.py,
// profiles/r02_ab_pass_edges.txt) the unique-edge form is 0.4 % SLOWER per pass than the records (90 vs 85.5 us per step): a
// line shares only the edge it crosses with the next cell, so folding leaves 3 entries per record, not 2 -- 48 B of stream
// instead of 40 B -- and the 64-B sectors the gathers pull are the same ones either way.
static int g_use_edges = 0;
void integral_use_edges(int on) { g_use_edges = on; }
int integral_uses_edges() { return g_use_edges; }

int launch_integral(const WeightSet &ws, const double *data, long ncell, int planes, long nx,
                    const int *tr_offsets_dev, int ntransect, double *scratch, double *row, hipStream_t s, int nsteps,
                    long data_stride, long row_stride)
{
    const unsigned ny = (unsigned)(nsteps > 1 ? nsteps : 1);
    if (planes == 2 && ws.ent_start && g_use_edges) {   // the engine's own planes through the unique-edge entries
        if (ws.nent > 0)
            hipLaunchKernelGGL(k_gather_edges, dim3((unsigned)((ws.nent + kBlock - 1) / kBlock), ny), dim3(kBlock), 0, s,
                               ws.ent, ws.nent, data, ncell, scratch, data_stride);
        if (ws.nseg > 0) {
            const unsigned nb = (unsigned)(((long)ws.nseg * kWave + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(k_finalize_seg, dim3(nb, ny), dim3(kBlock), 0, s, scratch, ws.ent_start, ws.nseg, row,
                               ws.nent, row_stride);
        }
        if (ntransect > 0) {
            const unsigned nb = (unsigned)(((long)ntransect * kWave + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(k_finalize_tr, dim3(nb, ny), dim3(kBlock), 0, s, tr_offsets_dev, ntransect, ws.nseg, row,
                               row_stride);
        }
        NF_HIP(hipGetLastError());
        return NF_OK;
    }
    if (ws.nrec > 0) {
        hipLaunchKernelGGL(k_gather_segscan, dim3((unsigned)((ws.nrec + kBlock - 1) / kBlock), ny), dim3(kBlock), 0, s,
                           ws.cell, ws.w4, ws.seg, ws.nrec, data, ncell, planes, (unsigned)(nx > 0 ? nx : 1), scratch,
                           data_stride);
    }
    if (ws.nseg > 0) {
        const unsigned nb = (unsigned)(((long)ws.nseg * kWave + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_finalize_seg, dim3(nb, ny), dim3(kBlock), 0, s, scratch, ws.seg_start, ws.nseg, row,
                           ws.nrec, row_stride);
    }
    if (ntransect > 0) {
        const unsigned nb = (unsigned)(((long)ntransect * kWave + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_finalize_tr, dim3(nb, ny), dim3(kBlock), 0, s, tr_offsets_dev, ntransect, ws.nseg, row,
                           row_stride);
    }
    NF_HIP(hipGetLastError());
    return NF_OK;
}

}  // namespace nf
