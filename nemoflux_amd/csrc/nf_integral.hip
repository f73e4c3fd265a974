// nf_integral.hip -- K3: gather of the weighted edge fluxes and their reduction onto target segments.
//
// Replaces  mint.PolylineIntegral.getIntegral(data, mint.CELL_BY_CELL_DATA) as driven by
//           nemoflux/field.py:102 and nemoflux/fluxplot.py:56, for ALL transects of a Field at once.
//
// Entries (cell*4+edge, weight, global segment id) are sorted by segment (K2).  Stage A: one lane per
// entry gathers data[cell,edge] (either the reference's (ncell,4) AoS or the engine's resident [4][ncell]
// planes), multiplies by the weight and runs a WAVEFRONT SEGMENTED SCAN keyed by the segment id (6
// shuffle steps); the last lane of every run inside the wave stores the run's sum.  Stage B: one wavefront
// per transect; each lane stitches the per-wave run sums of its segments (in wave order) into the
// per-segment total, then a butterfly adds the segments into the transect total.  No atomics: the
// summation tree is fixed, so results are bitwise reproducible.
//
// Output row: [ per-segment sums (nseg) | per-transect sums (ntransect) ].
#include "nf_common.h"

namespace nf {

__global__ __launch_bounds__(kBlock) void k_gather_segscan(const int64_t *__restrict__ cell_edge,
                                                           const double *__restrict__ weight,
                                                           const int *__restrict__ seg, long n,
                                                           const double *__restrict__ data, long ncell,
                                                           int planes, double *__restrict__ runsum)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    double val = 0.0;
    int key = -1;
    if (k < n) {
        const int64_t ce = cell_edge[k];
        const long addr = planes ? (long)(ce & 3) * ncell + (long)(ce >> 2) : (long)ce;
        val = weight[k] * data[addr];
        key = seg[k];
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

__global__ __launch_bounds__(kBlock) void k_finalize(const double *__restrict__ runsum,
                                                     const int *__restrict__ seg_start,
                                                     const int *__restrict__ tr_off, int ntransect, int nseg,
                                                     double *__restrict__ row)
{
    const int p = (blockIdx.x * kBlock + threadIdx.x) / kWave;  // one wavefront per transect
    const int lane = threadIdx.x & (kWave - 1);
    if (p >= ntransect) return;
    double part = 0.0;
    for (int s = tr_off[p] + lane; s < tr_off[p + 1]; s += kWave) {
        const long lo = seg_start[s], hi = seg_start[s + 1];
        double acc = 0.0;
        long first = lo;
        while (first < hi) {  // the run's pieces, one per wave it spans, in order
            long e = (first / kWave + 1) * kWave;
            if (e > hi) e = hi;
            acc += runsum[e - 1];
            first = e;
        }
        row[s] = acc;
        part += acc;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, kWave);
    if (lane == 0) row[nseg + p] = part;
}

int launch_integral(const WeightSet &ws, const double *data, long ncell, int planes, const int *tr_offsets_dev,
                    int ntransect, double *scratch, double *row, hipStream_t s)
{
    if (ws.n > 0) {
        hipLaunchKernelGGL(k_gather_segscan, dim3((unsigned)((ws.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           ws.cell_edge, ws.weight, ws.seg, ws.n, data, ncell, planes, scratch);
    }
    if (ntransect > 0) {
        const unsigned nb = (unsigned)(((long)ntransect * kWave + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_finalize, dim3(nb), dim3(kBlock), 0, s, scratch, ws.seg_start, tr_offsets_dev,
                           ntransect, ws.nseg, row);
    }
    NF_HIP(hipGetLastError());
    return NF_OK;
}

}  // namespace nf
