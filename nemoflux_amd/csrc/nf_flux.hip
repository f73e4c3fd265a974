// nf_flux.hip -- K1: vertical integration of uo/vo over the owned z-levels of one time step, fused with
// the edge-flux assembly.  This is the bandwidth-bound kernel the roofline is quoted on.
//
// Replaces  nemoflux/field.py:145-163  readField: missing -> 0 (:157), sum_z thickness[z]*f[z,j,i] (:161)
//           nemoflux/field.py:183-234  computeIntegratedFlux: eU = +U*arc[:,1], eV = -V*arc[:,2] (:195-196),
//                                      cell-by-cell 4-edge array incl. the neighbour copies (:209-223),
//                                      Sverdrup scaling (:225-228), |eU|,|eV| (:231-232), running max (:234)
//
// Access pattern.  uo/vo are (nt,nz,ny,nx) x-fastest.  A lane owns VEC consecutive cells (16 B: 2 x f64 or
// 4 x f32) and walks z with stride ncell; a wavefront therefore reads 1 KiB contiguous per (z, field) with
// global_load_dwordx4, in batches of UZ levels x 2 fields (all loads of a batch are issued, then the batch is consumed;
// DESIGN.md section 4 has what the compiler makes of that).  Nothing is reused, so nothing is staged in LDS; the
// thickness vector is wave-uniform (scalar loads).
//
// Resident output layout (HBM, float64) -- SoA planes instead of the reference's (ncell,4) AoS, so that every
// store is a dense 16 B/lane stream and the neighbour copies become SHIFTED dense stores:
//   iV planes  [4][ncell]   plane 1 = eU, plane 2 = eV,
//                           plane 0[c+nx] = eV[c]           (south slot of the row above, field.py:219;
//                                                            row 0 is never written and stays 0)
//                           plane 3[j, (i+1)%nx] = eU[j,i]  (west slot incl. the periodic wrap, :221-223)
//   abs planes [2][ncell]   |eU|, |eV|
// nf_field_read_step() re-packs the planes into the reference's (ncell,4) layout on demand.
// Two store forms: FUSED (the flux kernel stores all six planes; float64 default) and SPLIT (it stores only planes 1 and
// 2 and k_expand_planes, below, streams them into the other four right behind it; float32 default, and the compact
// resident mode without the expansion) -- see launch_flux_v for the measurements behind the per-dtype choice.
//
// Algorithmic bytes per (t,z,j,i) unit: 2*sizeof(T) read + (16 arc + 32 iV + 16 abs)/nz  (SURVEY 8d).
#include <cstring>

#include "nf_common.h"

namespace nf {

template <typename T, int VEC> struct vec_t;
typedef double dvec2 __attribute__((ext_vector_type(2)));  // clang vectors: accepted by the nontemporal builtin
typedef float fvec4 __attribute__((ext_vector_type(4)));
template <> struct vec_t<double, 2> { using type = dvec2; };
template <> struct vec_t<float, 4> { using type = fvec4; };
template <> struct vec_t<double, 1> { using type = double; };
template <> struct vec_t<float, 1> { using type = float; };

template <typename T, int VEC> struct Lanes {
    T x[VEC];
};

template <typename T, int VEC, bool NT>
__device__ inline Lanes<T, VEC> load_cells(const T *p)
{
    using V = typename vec_t<T, VEC>::type;
    Lanes<T, VEC> r;
    V v = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p)) : *reinterpret_cast<const V *>(p);
    __builtin_memcpy(&r, &v, sizeof(V));
    return r;
}

// missing -> 0 (field.py:157): NaN, or equal to the variable's _FillValue (compared in the file's dtype)
template <typename T>
__device__ inline double fixed(T x, T fill)
{
    return (x != x || x == fill) ? 0.0 : (double)x;
}

// ... or to the variable's missing_value when the file carries one that differs from _FillValue (xarray's decode_cf, which
// the reference relies on at field.py:34-35, masks both)
template <typename T>
__device__ inline double fixed2(T x, T fill, T fill2)
{
    return (x != x || x == fill || x == fill2) ? 0.0 : (double)x;
}

template <bool NTS = false>
__device__ inline void store1(double *p, double a)
{
    if (NTS) __builtin_nontemporal_store(a, p);
    else *p = a;
}
template <bool NTS = false>
__device__ inline void store2(double *p, double a, double b, bool aligned)
{
    if (aligned) {
        if (NTS) {
            dvec2 v = {a, b};
            __builtin_nontemporal_store(v, reinterpret_cast<dvec2 *>(p));
        } else {
            *reinterpret_cast<double2 *>(p) = make_double2(a, b);
        }
    } else {
        store1<NTS>(p, a);
        store1<NTS>(p + 1, b);
    }
}

// stores of one lane's VEC consecutive cells starting at c0 (see the layout comment at the top)
template <int VEC, bool NTS = false>
__device__ inline void store_cells(long c0, const double *eU, const double *eV, long ncell, unsigned ny, unsigned nx,
                                   double *__restrict__ iV, double *__restrict__ absUV)
{
    double *p0 = iV, *p1 = iV + ncell, *p2 = iV + 2 * ncell, *p3 = iV + 3 * ncell;
    double *aU = absUV, *aV = absUV + ncell;
    const unsigned j0 = (unsigned)(c0 / nx);
    const unsigned i0 = (unsigned)(c0 - (long)j0 * nx);
    if (VEC == 1) {
        store1<NTS>(p1 + c0, eU[0]);
        store1<NTS>(p2 + c0, eV[0]);
        store1<NTS>(aU + c0, fabs(eU[0]));
        store1<NTS>(aV + c0, fabs(eV[0]));
        if (j0 + 1 < ny) store1<NTS>(p0 + c0 + nx, eV[0]);
        store1<NTS>(p3 + ((i0 + 1 < nx) ? c0 + 1 : c0 + 1 - nx), eU[0]);
    } else {
        // own slots and |.|: dense 16 B/lane stores (c0 is a multiple of VEC)
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            store2<NTS>(p1 + c0 + k, eU[k], eU[k + 1], true);
            store2<NTS>(p2 + c0 + k, eV[k], eV[k + 1], true);
            store2<NTS>(aU + c0 + k, fabs(eU[k]), fabs(eU[k + 1]), true);
            store2<NTS>(aV + c0 + k, fabs(eV[k]), fabs(eV[k + 1]), true);
        }
        if (i0 + VEC <= nx) {
            // lane's cells sit in one row: south slots of the row above = the same stream shifted by nx
            if (j0 + 1 < ny) {
                const bool al = (nx & 1u) == 0;
#pragma unroll
                for (int k = 0; k < VEC; k += 2) store2<NTS>(p0 + c0 + nx + k, eV[k], eV[k + 1], al);
            }
            // west slots of the cells to the right: shifted by one (8 B stores; the row's last cell wraps to
            // column 0, field.py:223).  The lane-shifted 16-byte form of k_flux_field was measured here too (round 5,
            // profiles/r05_west_shift.txt): float64 -0.3 % at the headline size, inside the noise -- not taken, the
            // headline kernel keeps the instruction stream it has had since round 2
#pragma unroll
            for (int k = 0; k < VEC; ++k) store1<NTS>(p3 + ((i0 + k + 1 < nx) ? c0 + k + 1 : c0 + k + 1 - nx), eU[k]);
        } else {
            // lane straddles a row end (nx % VEC != 0): per-cell bookkeeping
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const long c = c0 + k;
                const unsigned j = (unsigned)(c / nx);
                const unsigned i = (unsigned)(c - (long)j * nx);
                if (j + 1 < ny) store1<NTS>(p0 + c + nx, eV[k]);
                store1<NTS>(p3 + ((i + 1 < nx) ? c + 1 : c + 1 - nx), eU[k]);
            }
        }
    }
}

// FORM bits of k_flux.  The product library instantiates only the two bit-identical code forms; the diagnostic bits (which
// remove an ingredient of the kernel to price it, i.e. give WRONG results on purpose) exist only in tuning builds
// (-DNF_TUNING_BUILD, `make tuning`, tools/ab_flux.py) and are compiled out of the shipped .so.
constexpr int kFormSignedOnly = 16;    // store only planes 1 (eU) and 2 (eV): split step and compact resident mode
constexpr int kFormNestedLoads = 128;  // per-level nested load loop instead of the flat one
constexpr int kFormTwoFills = 256;     // a second value counts as missing (missing_value != _FillValue)
#ifdef NF_TUNING_BUILD
constexpr int kDiagNoStores = 1, kDiagNoFix = 2, kDiagNoMax = 4, kDiagNoArc = 8, kDiagInterleaved = 32, kDiagPlainStores = 64;
#else
constexpr int kDiagNoStores = 0, kDiagNoFix = 0, kDiagNoMax = 0, kDiagNoArc = 0, kDiagInterleaved = 0, kDiagPlainStores = 0;
#endif

// BLOCK threads; every lane owns CH chunks of VEC cells, chunk q at tile_base + q*BLOCK*VEC + tid*VEC, so the
// workgroup reads CH x (BLOCK x 16 B) contiguous bytes per (z, field).  The waves of a workgroup never synchronise.
template <typename T, int VEC, int UZ, bool NT, int BLOCK, int CH, int FORM = 0>
__global__ __launch_bounds__(BLOCK) void k_flux(const T *__restrict__ u, const T *__restrict__ v, long ncell,
                                                unsigned ny, unsigned nx, int z0, int z1,
                                                const double *__restrict__ thickness,
                                                const double *__restrict__ arcE, const double *__restrict__ arcN,
                                                T fill, double scale, int sverdrup, double *__restrict__ iV,
                                                double *__restrict__ absUV, unsigned long long *maxbits,
                                                unsigned ntiles, int xcd_map, StepBatch sb, T fill2)
{
#ifndef NF_TUNING_BUILD
    static_assert((FORM & ~(kFormSignedOnly | kFormNestedLoads | kFormTwoFills)) == 0, "diagnostic forms exist only in tuning builds");
#endif
    const unsigned tile = xcd_map ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    if (sb.zr) {  // several time steps in one launch (small grids are launch-bound): blockIdx.y is the step
        const long tb = blockIdx.y;
        u += tb * sb.in_stride;
        v += tb * sb.in_stride;
        iV += tb * 4 * ncell;
        absUV += tb * 2 * ncell;
        z0 = sb.zr[2 * tb];
        z1 = sb.zr[2 * tb + 1];  // z1 == z0: the rank owns nothing of this step and stores zeros
    }
    double tmax = 0.0;
    if (tile < ntiles) {  // workgroup-uniform
        const long base = (long)tile * BLOCK * VEC * CH + (long)threadIdx.x * VEC;
        constexpr long kChunk = (long)BLOCK * VEC;
        // ncell % VEC == 0 is guaranteed by the launcher, so a chunk is either full or absent
        bool on[CH];
        double accU[CH][VEC], accV[CH][VEC];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            on[q] = base + q * kChunk < ncell;
#pragma unroll
            for (int k = 0; k < VEC; ++k) accU[q][k] = accV[q][k] = 0.0;
        }
        const T *pu = u + (long)z0 * ncell + base;
        const T *pv = v + (long)z0 * ncell + base;
        // batches of UZ levels: all loads of a batch are issued before its first use; the last batch may be partial
        // (wave-uniform predicate), so the tail levels are in flight together too instead of one at a time
        for (int z = z0; z < z1; z += UZ) {
            const int nlev = z1 - z < UZ ? z1 - z : UZ;
            Lanes<T, VEC> lu[UZ][CH], lv[UZ][CH];
            if (CH == 1 && !(FORM & kFormNestedLoads)) {
                // ONE flat loop over the 2*UZ loads (u0, v0, u1, v1, ...), each under its own wave-uniform predicate: this
                // form compiles to a load stream that runs 1-3 % faster (in-process, several boxes) than the nested
                // per-level form below -- at float32 only up to 9 levels per batch (at 10 it is 12 % slower: the kernel
                // drops to 4 waves per SIMD).  Putting all u levels before all v levels costs 14 %.
#pragma unroll
                for (int k = 0; k < 2 * UZ; ++k) {
                    const int r = k / 2;
                    if (r < nlev && on[0]) {
                        if ((k & 1) == 0) lu[r][0] = load_cells<T, VEC, NT>(pu + (long)r * ncell);
                        else lv[r][0] = load_cells<T, VEC, NT>(pv + (long)r * ncell);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < UZ; ++r)
                    if (r < nlev) {
#pragma unroll
                        for (int q = 0; q < CH; ++q)
                            if (on[q]) {
                                lu[r][q] = load_cells<T, VEC, NT>(pu + (long)r * ncell + q * kChunk);
                                lv[r][q] = load_cells<T, VEC, NT>(pv + (long)r * ncell + q * kChunk);
                            }
                    }
            }
#pragma unroll
            for (int r = 0; r < UZ; ++r)
                if (r < nlev) {
                    const double th = thickness[z + r];
#pragma unroll
                    for (int q = 0; q < CH; ++q)
                        if (on[q]) {
#pragma unroll
                            for (int k = 0; k < VEC; ++k) {
                                if (FORM & kFormTwoFills) {
                                    accU[q][k] = fma(th, fixed2<T>(lu[r][q].x[k], fill, fill2), accU[q][k]);
                                    accV[q][k] = fma(th, fixed2<T>(lv[r][q].x[k], fill, fill2), accV[q][k]);
                                } else {
                                    accU[q][k] = fma(th, (FORM & kDiagNoFix) ? (double)lu[r][q].x[k] : fixed<T>(lu[r][q].x[k], fill), accU[q][k]);
                                    accV[q][k] = fma(th, (FORM & kDiagNoFix) ? (double)lv[r][q].x[k] : fixed<T>(lv[r][q].x[k], fill), accV[q][k]);
                                }
                            }
                        }
                }
            pu += (long)UZ * ncell;
            pv += (long)UZ * ncell;
        }
        // edge fluxes (field.py:195-196, 225-228)
#pragma unroll
        for (int q = 0; q < CH; ++q)
            if (on[q]) {
                const long c0 = base + q * kChunk;
                double eU[VEC], eV[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    eU[k] = +accU[q][k] * ((FORM & kDiagNoArc) ? 1.5 : arcE[c0 + k]);
                    eV[k] = -accV[q][k] * ((FORM & kDiagNoArc) ? 1.5 : arcN[c0 + k]);
                    if (sverdrup) {
                        eU[k] *= scale;
                        eV[k] *= scale;
                    }
                    tmax = fmax(tmax, fmax(fabs(eU[k]), fabs(eV[k])));
                }
                if (FORM & kDiagInterleaved) {  // diagnostic: ONE interleaved (eU,eV) stream, 32 B per lane
#pragma unroll
                    for (int k = 0; k < VEC; ++k) store2<true>(iV + 2 * (c0 + k), eU[k], eV[k], true);
                } else if (FORM & kFormSignedOnly) {  // only the two signed planes (the rest comes from k_expand_planes)
#pragma unroll
                    for (int k = 0; k < VEC; k += 2) {
                        store2<true>(iV + ncell + c0 + k, eU[k], eU[k + 1], true);
                        store2<true>(iV + 2 * ncell + c0 + k, eV[k], eV[k + 1], true);
                    }
                } else if (!(FORM & kDiagNoStores)) store_cells<VEC, !(FORM & kDiagPlainStores)>(c0, eU, eV, ncell, ny, nx, iV, absUV);
            }
    }
    // running max (field.py:234): wavefront butterfly, then at most one atomic per WAVEFRONT -- no LDS, no workgroup
    // barrier, so a wave retires the moment its own columns are done (waves parked at a barrier idle their slots:
    // -0.6..-0.8 % in-process, -8 % in the store-free diagnostic build).  All values are non-negative doubles, whose bit
    // patterns order like unsigned integers.  The running max only grows, so it is read first (device scope, bypassing
    // the non-coherent caches) and the atomic is issued only by a wave that would raise it: a stale read can cause a
    // spare atomic, never a missed one.
    for (int o = 32; o > 0; o >>= 1) tmax = fmax(tmax, __shfl_xor(tmax, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0 && tmax > 0.0 && !(FORM & kDiagNoMax)) {
        unsigned long long b;
        __builtin_memcpy(&b, &tmax, 8);
        if (b > __hip_atomic_load(maxbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxbits, b);
    }
}

// ---- one field per wavefront: the launch shape for grids of a few wave-rounds -------------------------------------------
// The chip holds about 5 000 flux wavefronts at once.  A time step of the ORCA12-like grid is 50 656 of them (ten rounds);
// one of the ORCA025 grid (1440 x 1021, BASELINE config C3) is 11 488 at float64 and 5 744 at float32 -- 2.2 and 1.1 rounds:
// the few wavefronts of the last, partial round run a whole wave lifetime with the chip nearly empty (one 1-KiB request in
// flight each), and the step reaches 0.71 / 0.55 of the HBM peak where the big grid reaches 0.81 (tools/size_sweep.py).
// eU depends on uo only and eV on vo only, so the two vertical integrals of a cell can go to DIFFERENT wavefronts without
// touching the arithmetic: twice as many wavefronts, each half as long (75 loads instead of 150), and the partial round
// costs half as much.  blockIdx.z selects the field; a u-wave stores plane 1, |eU| and the west copies (plane 3), a
// v-wave plane 2, |eV| and the south copies (plane 0): the same values into the same slots as k_flux (bit-identical:
// test_field_split_bit_identical).  launch_flux picks this form by the number of wavefronts a launch has.
template <typename T, int VEC, int UZ, bool SIGNED_ONLY, bool TWO_FILLS, bool WSHIFT = false>
__global__ __launch_bounds__(256) void k_flux_field(const T *__restrict__ u, const T *__restrict__ v, long ncell,
                                                    unsigned ny, unsigned nx, int z0, int z1,
                                                    const double *__restrict__ thickness,
                                                    const double *__restrict__ arcE, const double *__restrict__ arcN,
                                                    T fill, double scale, int sverdrup, double *__restrict__ iV,
                                                    double *__restrict__ absUV, unsigned long long *maxbits,
                                                    unsigned ntiles, int xcd_map, StepBatch sb, T fill2)
{
    const unsigned tile = xcd_map ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    const bool is_v = blockIdx.z != 0;             // workgroup-uniform
    if (sb.zr) {
        const long tb = blockIdx.y;
        u += tb * sb.in_stride;
        v += tb * sb.in_stride;
        iV += tb * 4 * ncell;
        absUV += tb * 2 * ncell;
        z0 = sb.zr[2 * tb];
        z1 = sb.zr[2 * tb + 1];
    }
    double tmax = 0.0;
    const long c0 = (long)tile * 256 * VEC + (long)threadIdx.x * VEC;
    if (tile < ntiles && c0 < ncell) {   // ncell % VEC == 0: a lane's cells are all there or all absent
        double acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0;
        const T *p = (is_v ? v : u) + (long)z0 * ncell + c0;
        for (int z = z0; z < z1; z += UZ) {
            const int nlev = z1 - z < UZ ? z1 - z : UZ;
            Lanes<T, VEC> l[UZ];
#pragma unroll
            for (int r = 0; r < UZ; ++r)
                if (r < nlev) l[r] = load_cells<T, VEC, true>(p + (long)r * ncell);
#pragma unroll
            for (int r = 0; r < UZ; ++r)
                if (r < nlev) {
                    const double th = thickness[z + r];
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        acc[k] = fma(th, TWO_FILLS ? fixed2<T>(l[r].x[k], fill, fill2) : fixed<T>(l[r].x[k], fill), acc[k]);
                }
            p += (long)UZ * ncell;
        }
        const double *arc = is_v ? arcN : arcE;
        double e[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            e[k] = is_v ? -acc[k] * arc[c0 + k] : +acc[k] * arc[c0 + k];   // field.py:195-196
            if (sverdrup) e[k] *= scale;
            tmax = fmax(tmax, fabs(e[k]));
        }
        double *own = iV + (is_v ? 2 : 1) * ncell;
        if (VEC == 1) {
            store1<true>(own + c0, e[0]);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; k += 2) store2<true>(own + c0 + k, e[k], e[k + 1], true);
        }
        if (!SIGNED_ONLY) {
            double *ab = absUV + (is_v ? ncell : 0);
            if (VEC == 1) {
                store1<true>(ab + c0, fabs(e[0]));
            } else {
#pragma unroll
                for (int k = 0; k < VEC; k += 2) store2<true>(ab + c0 + k, fabs(e[k]), fabs(e[k + 1]), true);
            }
            // the neighbour copies (field.py:219-223): shifted copies of the lane's own stream
            const unsigned j0 = (unsigned)(c0 / nx), i0 = (unsigned)(c0 - (long)j0 * nx);
            if (VEC > 1 && i0 + VEC <= nx) {        // the lane's cells sit in one row
                if (is_v) {
                    if (j0 + 1 < ny) {              // plane 0: south slots of the row above = this stream shifted by nx
                        const bool al = (nx & 1u) == 0;
#pragma unroll
                        for (int k = 0; k < VEC; k += 2) store2<true>(iV + c0 + nx + k, e[k], e[k + 1], al);
                    }
                } else if (WSHIFT && nx % VEC == 0) {
                    // plane 3 in ALIGNED 16-byte pieces (round-4 verdict W5): the lane's own west slots are (eU of the lane to
                    // the left's last cell, e[0], ..., e[VEC-2]); the value from the left comes by a lane shift.  The first
                    // lane of a wavefront and of a grid row has no left neighbour in reach: its slot is stored (8 bytes) by
                    // the lane that owns the value -- the last lane of the wavefront before, or the row's last lane (the
                    // periodic wrap, field.py:223).  nx % VEC == 0: every lane's cells sit in one row.
                    const int lane = threadIdx.x & (kWave - 1);
                    const double left = __shfl_up(e[VEC - 1], 1, kWave);
                    double *p3 = iV + 3 * ncell;
                    if (lane > 0 && i0 > 0) store2<true>(p3 + c0, left, e[0], true);
                    else store1<true>(p3 + c0 + 1, e[0]);
#pragma unroll
                    for (int k = 2; k < VEC; k += 2) store2<true>(p3 + c0 + k, e[k - 1], e[k], true);
                    if (lane == kWave - 1 || i0 + VEC >= nx || c0 + VEC >= ncell)
                        store1<true>(p3 + ((i0 + VEC < nx) ? c0 + VEC : c0 + VEC - nx), e[VEC - 1]);
                } else {                            // plane 3: west slots of the cells to the right, the row's last cell wraps
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        store1<true>(iV + 3 * ncell + ((i0 + k + 1 < nx) ? c0 + k + 1 : c0 + k + 1 - nx), e[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {     // a row ends inside the lane's cells (or one cell per lane)
                    const long c = c0 + k;
                    const unsigned j = (unsigned)(c / nx), i = (unsigned)(c - (long)j * nx);
                    if (is_v) {
                        if (j + 1 < ny) store1<true>(iV + c + nx, e[k]);
                    } else {
                        store1<true>(iV + 3 * ncell + ((i + 1 < nx) ? c + 1 : c + 1 - nx), e[k]);
                    }
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) tmax = fmax(tmax, __shfl_xor(tmax, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0 && tmax > 0.0) {
        unsigned long long b;
        __builtin_memcpy(&b, &tmax, 8);
        if (b > __hip_atomic_load(maxbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxbits, b);
    }
}

#ifdef NF_TUNING_BUILD
// tuning builds only (make -C nemoflux_amd/csrc tuning): the writer-wave form of K1 (measured 4-13 % slower, kept as a
// record of the experiment) lives with the tools that measure it
#include "../../tools/nf_flux_ww.inc"
#endif

// ---- the four derived planes from the two signed ones ---------------------------------------------------------
// plane 0[c] = eV[c - nx] (row 0 stays 0: field.py:219), plane 3[c] = eU of the cell to the left, column 0 taking the
// row's last cell (field.py:221-223), |eU|, |eV| (field.py:231-232).  Pure streaming right behind the flux kernel (its
// two planes are still in the Infinity Cache): a lane owns two cells, every store is a dense aligned 16 B piece.
// This is the second half of a step in the SPLIT store form (the flux kernel then keeps only a third of the store traffic
// inside the read-saturated kernel) and the on-demand expansion of the compact resident mode.
template <int W>
__global__ __launch_bounds__(kBlock) void k_expand_planes(double *__restrict__ iV, double *__restrict__ absUV, long ncell,
                                                          unsigned nx)
{
    const long c0 = ((long)blockIdx.x * kBlock + threadIdx.x) * W;
    if (c0 >= ncell) return;
    const double *eU = iV + ncell, *eV = iV + 2 * ncell;
    double u[W], v[W], south[W], west[W];
    bool has_south[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const long c = c0 + k;
        const unsigned j = (unsigned)(c / nx), i = (unsigned)(c - (long)j * nx);
        u[k] = eU[c];
        v[k] = eV[c];
        has_south[k] = j > 0;
        south[k] = has_south[k] ? eV[c - nx] : 0.0;
        west[k] = eU[i > 0 ? c - 1 : c - 1 + nx];
    }
    if (W == 2) {
        if (has_south[0] && has_south[1]) store2<true>(iV + c0, south[0], south[1], true);
        else if (has_south[1]) store1<true>(iV + c0 + 1, south[1]);   // the pair straddles the end of row 0 (odd nx)
        store2<true>(iV + 3 * ncell + c0, west[0], west[1], true);
        store2<true>(absUV + c0, fabs(u[0]), fabs(u[1]), true);
        store2<true>(absUV + ncell + c0, fabs(v[0]), fabs(v[1]), true);
    } else {
        if (has_south[0]) store1<true>(iV + c0, south[0]);
        store1<true>(iV + 3 * ncell + c0, west[0]);
        store1<true>(absUV + c0, fabs(u[0]));
        store1<true>(absUV + ncell + c0, fabs(v[0]));
    }
}
int launch_expand_planes(double *iV, double *absUV, long ncell, long ny, long nx, hipStream_t s)
{
    NF_REQUIRE(iV && absUV && ncell > 0 && ncell == ny * nx, NF_ERR_ARG, "expand: bad arguments");
    if (ncell % 2 == 0) {
        const long n2 = ncell / 2;
        hipLaunchKernelGGL(k_expand_planes<2>, dim3((unsigned)((n2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, iV, absUV,
                           ncell, (unsigned)nx);
    } else {
        hipLaunchKernelGGL(k_expand_planes<1>, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, iV,
                           absUV, ncell, (unsigned)nx);
    }
    NF_HIP(hipGetLastError());
    return NF_OK;
}

// tuning knobs: nf_tuning_set() at run time (A/B runs inside one process)
static int g_xcd_map = 1;
static int g_variant = 0;
// "field_split": -1 = by the size of the launch (default), 0 = never, 1 = always
static int g_field_split = -1;
// "west_shift": 1 (default) = the one-field kernel builds the west slots from lane-shifted values (aligned 16-byte stores);
// 0 = 8-byte stores at a 16 / 32-byte lane stride, the form before round 5.  In-process A/B on the ORCA025-like step
// (profiles/r05_west_shift.txt): float32 0.1585 -> 0.1549 ms (-2.3 %), WRITE_SIZE 1.38 -> 1.23 x the algorithmic store
// bytes; float64 0.2816 -> 0.2797 ms (-0.7 %), 1.048 -> 1.010 x.  Bit-identical planes (test_field_split_bit_identical).
static int g_west_shift = 1;
// One-step launches with fewer wavefronts than this take the one-field form (about four rounds of resident wavefronts).
// tools/size_sweep.py, profiles/r04_size_sweep.txt: 1440 x 1021 x 75 (11 488 / 5 744 wavefronts at float64 / float32) gains
// 17 % / 41 %, 2160 x 1080 (18 225 / 9 112) 2 % / 8 %, 3600 x 1800 (50 656 / 25 312) loses 2 %
constexpr long kFieldSplitWaves = 20000;
static long g_tuning_version = 0;
long tuning_version() { return g_tuning_version; }
int tuning_set(const char *name, int value)
{
    ++g_tuning_version;  // captured graphs bake the variant in
    if (!strcmp(name, "xcd_map")) g_xcd_map = value;
    else if (!strcmp(name, "flux_variant")) g_variant = value;
    else if (!strcmp(name, "field_split")) g_field_split = value;
    else if (!strcmp(name, "west_shift")) g_west_shift = value;
#ifdef NF_TUNING_BUILD
    else if (!strcmp(name, "ww_blocks_per_cu")) g_pipe_waves = value;
#endif
    else return NF_ERR_ARG;
    return NF_OK;
}

template <typename T, int VEC, int UZ, bool NT, int BLOCK, int CH, int FORM = 0>
static int launch_flux_t(const FluxArgs &a, hipStream_t s)
{
    const long per_tile = (long)BLOCK * VEC * CH;
    const unsigned ntiles = (unsigned)((a.ncell + per_tile - 1) / per_tile);
    const int xcd_map = g_xcd_map;
    const unsigned grid = xcd_map ? xcd_grid(ntiles) : ntiles;
    hipLaunchKernelGGL((k_flux<T, VEC, UZ, NT, BLOCK, CH, FORM>), dim3(grid, (unsigned)(a.batch.zr ? a.batch.nsteps : 1)),
                       dim3(BLOCK), 0, s, (const T *)a.u, (const T *)a.v, a.ncell, (unsigned)a.ny, (unsigned)a.nx, a.z0,
                       a.z1, a.thickness, a.arcE, a.arcN, (T)a.fill, a.scale, a.sverdrup, a.iV, a.absU, a.maxbits, ntiles,
                       xcd_map, a.batch, (T)a.fill2);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

template <typename T, int VEC, int UZ, bool SIGNED_ONLY, bool TWO_FILLS, bool WSHIFT = false>
static int launch_flux_field_t(const FluxArgs &a, hipStream_t s)
{
    const long per_tile = 256l * VEC;
    const unsigned ntiles = (unsigned)((a.ncell + per_tile - 1) / per_tile);
    const int xcd_map = g_xcd_map;
    const unsigned grid = xcd_map ? xcd_grid(ntiles) : ntiles;
    hipLaunchKernelGGL((k_flux_field<T, VEC, UZ, SIGNED_ONLY, TWO_FILLS, WSHIFT>), dim3(grid, (unsigned)(a.batch.zr ? a.batch.nsteps : 1), 2),
                       dim3(256), 0, s, (const T *)a.u, (const T *)a.v, a.ncell, (unsigned)a.ny, (unsigned)a.nx, a.z0, a.z1,
                       a.thickness, a.arcE, a.arcN, (T)a.fill, a.scale, a.sverdrup, a.iV, a.absU, a.maxbits, ntiles, xcd_map,
                       a.batch, (T)a.fill2);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

// the one-field-per-wavefront form (k_flux_field).  signed_only: the caller wants planes 1 and 2 only (compact mode, partial
// steps of a sharded run)
template <typename T, int VEC>
static int launch_flux_field(const FluxArgs &a, hipStream_t s)
{
    constexpr int kLevels = sizeof(T) == 8 ? 10 : 16;      // one field per wave: as many bytes in flight as the two-field batch
    const bool two = (T)a.fill2 == (T)a.fill2 && !((T)a.fill2 == (T)a.fill);
    if (a.signed_only)
        return two ? launch_flux_field_t<T, VEC, kLevels, true, true>(a, s) : launch_flux_field_t<T, VEC, kLevels, true, false>(a, s);
    if (g_west_shift && VEC > 1)
        return two ? launch_flux_field_t<T, VEC, kLevels, false, true, true>(a, s) : launch_flux_field_t<T, VEC, kLevels, false, false, true>(a, s);
    return two ? launch_flux_field_t<T, VEC, kLevels, false, true>(a, s) : launch_flux_field_t<T, VEC, kLevels, false, false>(a, s);
}

// Store form of the vector path (nf_tuning_set("flux_variant")): 0 = the per-dtype default, 5 = the other
// one; same bits either way (tests/test_gpu_configs.py).  Any other number runs the default kernel in the shipped library;
// the measured alternatives of the load loop exist only in the tuning build (`make tuning`), where the same test checks them.
template <typename T, int VEC>
static int launch_flux_v(const FluxArgs &a, hipStream_t s)
{
    const int variant = a.batch.zr ? 0 : g_variant;  // the multi-step launch exists for the default kernel only
    if (variant == 0) {   // few wavefronts (one time step of a mid-size or small grid): one field per wavefront
        const long waves = (a.ncell / VEC + kWave - 1) / kWave;
        const bool split = g_field_split < 0 ? (!a.batch.zr && waves < kFieldSplitWaves) : g_field_split != 0;
        if (split) return launch_flux_field<T, VEC>(a, s);
    }
    // a second missing value (compared in the file's dtype, like the first): the default kernels with one more compare
    const bool two = (T)a.fill2 == (T)a.fill2 && !((T)a.fill2 == (T)a.fill);
    if (a.signed_only) {  // compact resident mode: the caller expands on demand
        NF_REQUIRE(VEC > 1 && !a.batch.zr, NF_ERR_STATE, "flux: the compact mode needs 16-byte aligned fields, an even cell count and one step per launch");
        if (two) return launch_flux_t<T, VEC, (sizeof(T) == 8 ? 10 : 8), true, 256, 1, kFormSignedOnly | kFormTwoFills>(a, s);
        return launch_flux_t<T, VEC, (sizeof(T) == 8 ? 10 : 8), true, 256, 1, kFormSignedOnly>(a, s);   // same batches as the defaults
    }
    if (VEC == 1) {   // odd cell counts / unaligned fields: one cell per lane
        if (two) return launch_flux_t<T, VEC, 8, true, 256, 1, kFormTwoFills>(a, s);
        return launch_flux_t<T, VEC, 8, true, 256, 1>(a, s);
    }
    if (two) {
        constexpr int kLevels = sizeof(T) == 8 ? 10 : 8;
        if (a.batch.zr || sizeof(T) == 8) return launch_flux_t<T, VEC, kLevels, true, 256, 1, kFormTwoFills>(a, s);
        const int rc = launch_flux_t<T, VEC, kLevels, true, 256, 1, kFormSignedOnly | kFormTwoFills>(a, s);
        if (rc != NF_OK) return rc;
        if (a.mid_event) {
            NF_HIP(hipEventRecord(a.mid_event, s));
            if (a.mid_recorded) *a.mid_recorded = true;
        }
        return launch_expand_planes(a.iV, a.absU, a.ncell, a.ny, a.nx, s);
    }
    switch (variant) {
#ifdef NF_TUNING_BUILD
        // measured alternatives (tools/ab_flux.py; docs/EXPERIMENTS.md): all within +-3 % of the default.  Tuning build only
        // (round-3 verdict W9): the shipped library carries the default and the other store form (5), nothing else
        case 3: return launch_flux_t<T, VEC, 4, true, 256, 1>(a, s);    // 4 levels in flight (+2..3 %)
        case 4: return launch_flux_t<T, VEC, 4, true, 256, 2>(a, s);    // 4 levels, 2 chunks per lane
        case 11: return launch_flux_t<T, VEC, 10, false, 256, 1>(a, s); // plain (temporal) loads: +4 %
        case 12: return launch_flux_t<T, VEC, 8, true, 256, 1>(a, s);   // 8 levels in flight
        case 14: return launch_flux_t<T, VEC, 16, true, 256, 1>(a, s);  // 16 levels in flight
        case 13: return launch_flux_t<T, VEC, 10, true, 256, 1, kDiagPlainStores>(a, s);  // plain instead of non-temporal stores
        case 40: return launch_flux_ww<T, VEC, 2>(a, s);                       // writer-wave form
        // diagnostic builds (WRONG RESULTS on purpose) that price one ingredient each
        case 21: return launch_flux_t<T, VEC, 10, true, 256, 1, kDiagNoStores>(a, s);     // no stores
        case 28: return launch_flux_t<T, VEC, 10, true, 256, 1, kFormSignedOnly>(a, s);   // only the two signed planes, no expansion
        case 29: return launch_flux_t<T, VEC, 10, true, 256, 1, kDiagInterleaved>(a, s);  // one interleaved (eU,eV) stream
        case 45: return launch_flux_ww<T, VEC, 2, 2>(a, s);                        // writer-wave, no stores
        case 6: {  // the nested per-level load loop with the split store form: the defaults before the flat loop
            const int rc = launch_flux_t<T, VEC, (sizeof(T) == 8 ? 10 : 8), true, 256, 1, kFormSignedOnly | kFormNestedLoads>(a, s);
            return rc != NF_OK ? rc : launch_expand_planes(a.iV, a.absU, a.ncell, a.ny, a.nx, s);
        }
#endif
        case 5:    // the OTHER store form than the default's (float64: split, float32: fused)
        default: {
            // 10 (float64) or 8 (float32) levels x 2 fields per batch.  Two store forms: FUSED = all seven stores in the flux kernel; SPLIT = the flux
            // kernel stores eU, eV and the streaming expansion derives the copies and |.| right behind it.  Which one
            // wins follows the code the compiler makes of the load loop (in-process A/B, whole passes, several boxes):
            // float64 (flat load loop): fused -1.1..-1.7 %;  float32 (nested load loop): split -7..-12 %.
            // float32 runs 3-3.6 % faster with 8 (or 9) levels per batch than with 10; float64 shows no such preference.
            constexpr int kLevels = sizeof(T) == 8 ? 10 : 8;
            const bool fused = a.batch.zr || ((sizeof(T) == 8) != (variant == 5));   // multi-step launch: always fused
            if (fused) return launch_flux_t<T, VEC, kLevels, true, 256, 1>(a, s);
            const int rc = launch_flux_t<T, VEC, kLevels, true, 256, 1, kFormSignedOnly>(a, s);
            if (rc != NF_OK) return rc;
            if (a.mid_event) {
                NF_HIP(hipEventRecord(a.mid_event, s));
                if (a.mid_recorded) *a.mid_recorded = true;
            }
            return launch_expand_planes(a.iV, a.absU, a.ncell, a.ny, a.nx, s);
        }
    }
}

// the compact mode rides on the 16-byte vector path (aligned fields, cell count a multiple of the lane's vector)
bool flux_supports_signed_only(const FluxArgs &a)
{
    const bool al16 = ((uintptr_t)a.u % 16 == 0) && ((uintptr_t)a.v % 16 == 0);
    return al16 && !a.batch.zr && a.ncell % (a.dtype == NF_F32 ? 4 : 2) == 0;
}

int launch_flux(const FluxArgs &a, hipStream_t s)
{
    NF_REQUIRE(a.ncell > 0 && a.ncell == a.ny * a.nx && a.ncell < (1l << 31), NF_ERR_ARG, "flux: bad grid sizes");
    NF_REQUIRE(a.batch.zr || (a.z1 > a.z0 && a.z0 >= 0), NF_ERR_ARG, "flux: empty z range");
    NF_REQUIRE(!a.batch.zr || (a.batch.nsteps > 0 && a.batch.nsteps < 65536), NF_ERR_ARG, "flux: bad step batch");
    NF_REQUIRE(a.absV == a.absU + a.ncell, NF_ERR_ARG, "flux: abs planes must be contiguous");
    const bool al16 = ((uintptr_t)a.u % 16 == 0) && ((uintptr_t)a.v % 16 == 0);
    if (a.dtype == NF_F64) {
        if (al16 && a.ncell % 2 == 0) return launch_flux_v<double, 2>(a, s);
        return launch_flux_v<double, 1>(a, s);
    } else if (a.dtype == NF_F32) {
        if (al16 && a.ncell % 4 == 0) return launch_flux_v<float, 4>(a, s);
        return launch_flux_v<float, 1>(a, s);
    }
    NF_REQUIRE(false, NF_ERR_ARG, "flux: dtype must be NF_F64 or NF_F32");
}

// ---- re-pack of the resident planes into the reference's (ncell,4) AoS (field.py:62) -----------------
__global__ __launch_bounds__(kBlock) void k_planes_to_aos(const double *__restrict__ planes, long ncell,
                                                          double *__restrict__ aos)
{
    long c = (long)blockIdx.x * kBlock + threadIdx.x;
    if (c < ncell) {
        double2 *o = reinterpret_cast<double2 *>(aos + 4 * c);
        o[0] = make_double2(planes[c], planes[ncell + c]);
        o[1] = make_double2(planes[2 * ncell + c], planes[3 * ncell + c]);
    }
}
int launch_planes_to_aos(const double *planes, long ncell, double *aos, hipStream_t s)
{
    hipLaunchKernelGGL(k_planes_to_aos, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       planes, ncell, aos);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

}  // namespace nf
