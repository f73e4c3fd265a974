// lds_chain.hip -- what does one step of a SERIAL decoder cost a lone wavefront on gfx950?  (tuning aid, not product)
//
// The DEFLATE decoder of nf_inflate_core.h is a chain of dependent LDS table lookups run by one wavefront per stream, four
// streams per CU.  This probe times the ingredients of such a chain in isolation -- one workgroup of 64 lanes per CU slot,
// N dependent iterations, hipEvent time / N -- so that the decoder's design can be priced against the hardware instead of
// against instruction counts:
//   lds_scalar    index in an SGPR -> v_mov -> ds_read_b32 -> s_waitcnt -> v_readfirstlane -> s_and   (the decoder's lookup)
//   lds_vector    index in a VGPR -> ds_read_b32 -> s_waitcnt -> v_and                               (all-VALU chain)
//   +salu8 / +valu8   the same with 8 more dependent SALU / VALU instructions in the chain
//   +branch2      the same with two taken s_branch hops per iteration
//   readlane      v_readlane (lane select in an SGPR) -> s_add -> next lane select: the hop of a register-resident chain
//   register table   the same chase through a table held in 16 VGPRs (VGPR index mode + v_readlane): what the decoder ships
//   literal step ... the decoder's literal iteration as shipped, and with one ingredient removed or replaced at a time
//                    (modes 20-32): what a lone wavefront pays for mixing scalar and vector instructions in a dependent loop
// Build / run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/lds_chain.hip -o build/lds_chain && build/lds_chain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

constexpr int kTab = 1024;

template <int MODE>
__global__ __launch_bounds__(64) void k_chain(const unsigned *init, int iters, unsigned *out)
{
    __shared__ unsigned tab[2 * kTab];    // second half: target of the byte stores of mode 8
    for (int k = threadIdx.x; k < kTab; k += 64) tab[k] = init[k];
    __syncthreads();
    unsigned idx = 0;
    if (MODE == 0) {            // lds_scalar
        for (int i = 0; i < iters; ++i)
            asm volatile("v_mov_b32 v40, %0\n\tds_read_b32 v41, v40\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, v41\n\t"
                         "s_and_b32 %0, %0, 0xffc"
                         : "+s"(idx)::"v40", "v41", "memory");
    } else if (MODE == 1) {     // lds_scalar + 8 SALU
        for (int i = 0; i < iters; ++i)
            asm volatile("v_mov_b32 v40, %0\n\tds_read_b32 v41, v40\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, v41\n\t"
                         "s_and_b32 %0, %0, 0xffc\n\ts_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 4\n\ts_add_u32 %0, %0, 8\n\t"
                         "s_sub_u32 %0, %0, 8\n\ts_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 4\n\ts_add_u32 %0, %0, 8\n\ts_sub_u32 %0, %0, 8"
                         : "+s"(idx)::"v40", "v41", "scc", "memory");
    } else if (MODE == 2) {     // lds_scalar + two taken branches
        for (int i = 0; i < iters; ++i)
            asm volatile("v_mov_b32 v40, %0\n\tds_read_b32 v41, v40\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, v41\n\t"
                         "s_and_b32 %0, %0, 0xffc\n\ts_branch 1f\n2:\n\ts_branch 3f\n1:\n\ts_branch 2b\n3:"
                         : "+s"(idx)::"v40", "v41", "memory");
    } else if (MODE == 3) {     // lds_vector
        unsigned v = 0;
        for (int i = 0; i < iters; ++i)
            asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xffc, %0" : "+v"(v)::"memory");
        idx = v;
    } else if (MODE == 4) {     // lds_vector + 8 VALU
        unsigned v = 0;
        for (int i = 0; i < iters; ++i)
            asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xffc, %0\n\tv_add_u32 %0, 4, %0\n\t"
                         "v_subrev_u32 %0, 4, %0\n\tv_add_u32 %0, 8, %0\n\tv_subrev_u32 %0, 8, %0\n\tv_add_u32 %0, 4, %0\n\t"
                         "v_subrev_u32 %0, 4, %0\n\tv_add_u32 %0, 8, %0\n\tv_subrev_u32 %0, 8, %0"
                         : "+v"(v)::"memory");
        idx = v;
    } else if (MODE == 5) {     // readlane hop: lane select from the previous hop
        unsigned v = (threadIdx.x * 7 + 3) & 63;     // a permutation of the lanes
        unsigned s = 0;
        for (int i = 0; i < iters; ++i)
            asm volatile("s_nop 4\n\tv_readlane_b32 %0, %1, %0\n\ts_and_b32 %0, %0, 63" : "+s"(s) : "v"(v));
        idx = s;
    } else if (MODE == 6) {     // SALU only: 16 dependent adds
        for (int i = 0; i < iters; ++i)
            asm volatile("s_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3\n\t"
                         "s_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3\n\t"
                         "s_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3\n\t"
                         "s_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 4\n\ts_sub_u32 %0, %0, 3"
                         : "+s"(idx)::"scc");
    } else if (MODE == 7) {     // VALU only: 16 dependent adds
        unsigned v = threadIdx.x;
        for (int i = 0; i < iters; ++i)
            asm volatile("v_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0\n\tv_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0\n\t"
                         "v_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0\n\tv_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0\n\t"
                         "v_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0\n\tv_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0\n\t"
                         "v_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0\n\tv_add_u32 %0, 4, %0\n\tv_subrev_u32 %0, 3, %0"
                         : "+v"(v));
        idx = v;
    } else if (MODE == 8) {     // the decoder's literal step, verbatim in shape (lookup, byte store, exit test), on a table of literals
        unsigned long long buf = 0x123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        for (int i = 0; i < iters; ++i)
            asm volatile("s_and_b32 s47, %1, 15\n\ts_lshr_b64 %0, %0, s47\n\ts_lshl_b32 s47, %2, 2\n\tv_mov_b32 v40, s47\n\t"
                         "v_and_b32 v40, 0xffc, v40\n\tds_read_b32 v41, v40\n\tv_mov_b32 v42, %2\n\tv_and_b32 v42, 0xfff, v42\n\t"
                         "v_mov_b32 v44, %1\n\tds_write_b8 v42, v44 offset:4096\n\ts_add_u32 %2, %2, 1\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "v_readfirstlane_b32 %1, v41\n\ts_or_b32 %1, %1, 5"
                         : "+s"(buf), "+s"(e), "+s"(pos)::"s47", "v40", "v41", "v42", "v44", "scc", "memory");
        idx = e + pos;
    }
    else if (MODE == 11) {      // mode 8 waiting only for the LOOKUP (lgkmcnt(1)): the younger byte store stays in flight
        unsigned long long buf = 0x123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        for (int i = 0; i < iters; ++i)
            asm volatile("s_and_b32 s47, %1, 15\n\ts_lshr_b64 %0, %0, s47\n\ts_lshl_b32 s47, %2, 2\n\tv_mov_b32 v40, s47\n\t"
                         "v_and_b32 v40, 0xffc, v40\n\tds_read_b32 v41, v40\n\tv_mov_b32 v42, %2\n\tv_and_b32 v42, 0xfff, v42\n\t"
                         "v_mov_b32 v44, %1\n\tds_write_b8 v42, v44 offset:4096\n\ts_add_u32 %2, %2, 1\n\ts_waitcnt lgkmcnt(1)\n\t"
                         "v_readfirstlane_b32 %1, v41\n\ts_or_b32 %1, %1, 5"
                         : "+s"(buf), "+s"(e), "+s"(pos)::"s47", "v40", "v41", "v42", "v44", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 9) {       // the same chase as lds_scalar through a REGISTER-resident table: 1024 entries = 16 VGPRs x 64 lanes,
                                // row picked by VGPR index mode (s_set_gpr_idx_on), lane by v_readlane -- no LDS in the chain
        asm volatile("v_lshlrev_b32 v40, 2, %1\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "s_mov_b32 s48, %2\n"
                     "1:\n\t"
                     "s_lshr_b32 s46, %0, 8\n\t"              // idx is a byte offset: row = idx >> 8, lane = (idx >> 2) & 63
                     "s_bfe_u32 s47, %0, 0x60002\n\t"
                     "s_set_gpr_idx_on s46, 0x1\n\t"
                     "v_mov_b32 v41, v60\n\t"
                     "s_set_gpr_idx_off\n\t"
                     "v_readlane_b32 %0, v41, s47\n\t"
                     "s_and_b32 %0, %0, 0xffc\n\t"
                     "s_sub_u32 s48, s48, 1\n\t"
                     "s_cmp_lg_u32 s48, 0\n\t"
                     "s_cbranch_scc1 1b"
                     : "+s"(idx)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s46", "s47", "s48", "v40", "v41", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70",
                       "v71", "v72", "v73", "v74", "v75", "scc", "memory");
    } else if (MODE == 10) {    // lds_scalar with the loop inside the asm (like mode 9): the like-for-like partner
        asm volatile("s_mov_b32 s48, %1\n"
                     "1:\n\t"
                     "v_mov_b32 v40, %0\n\tds_read_b32 v41, v40\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, v41\n\t"
                     "s_and_b32 %0, %0, 0xffc\n\t"
                     "s_sub_u32 s48, s48, 1\n\ts_cmp_lg_u32 s48, 0\n\ts_cbranch_scc1 1b"
                     : "+s"(idx) : "s"(iters) : "s48", "v40", "v41", "scc", "memory");
    } else if (MODE == 20) {    // the decoder literal step as shipped (register table, exec_lo from the entry, one ds_write_b8)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, %2, %3\n\tv_and_b32 v42, 0xfff, v42\n\tv_bfe_u32 v44, %1, v48, 8\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 33) {    // the shipped step re-ordered: every scalar instruction first, then ONE cluster of vector instructions (store + lookup)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\ts_mov_b32 s38, %2\n\ts_mov_b32 s37, %1\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_add_u32 v42, s38, %3\n\tv_and_b32 v42, 0xfff, v42\n\tv_bfe_u32 v44, s37, v48, 8\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s37", "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 21) {    // the same without the exec write (both lanes always store)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\tv_add_u32 v42, %2, %3\n\tv_and_b32 v42, 0xfff, v42\n\tv_bfe_u32 v44, %1, v48, 8\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 22) {    // the same without the store
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 23) {    // the same without the lookup (entry constant)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     ""
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, %2, %3\n\tv_and_b32 v42, 0xfff, v42\n\tv_bfe_u32 v44, %1, v48, 8\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "s_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 31) {    // no lookup AND VGPR index mode never switched on: the plain cost of the store sequence
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_nop 0\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     ""
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, %2, %3\n\tv_and_b32 v42, 0xfff, v42\n\tv_bfe_u32 v44, %1, v48, 8\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "s_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_nop 0\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 32) {    // index mode switched on only around the lookup (on, v_mov, off, v_readlane); the store runs in normal mode
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_nop 0\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_mov_b64 exec, -1\n\ts_set_gpr_idx_on s60, 0x1\n\tv_mov_b32 v41, v60\n\ts_set_gpr_idx_off\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, %2, %3\n\tv_and_b32 v42, 0xfff, v42\n\tv_bfe_u32 v44, %1, v48, 8\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v41, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_nop 0\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v41", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 24) {    // the shipped step with the store reduced to the ds_write_b8 (constant address and data)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 25) {    // the shipped step with the store reduced to its three vector instructions (no ds_write_b8)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, %2, %3\n\tv_and_b32 v42, 0xfff, v42\n\tv_bfe_u32 v44, %1, v48, 8\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 27) {    // vector part = v_add (position-dependent) only
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, %2, %3\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 28) {    // vector part = v_bfe (entry-dependent) only
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_bfe_u32 v44, %1, v48, 8\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 29) {    // vector part = s_lshr + v_mov of the entry (SALU copy first)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\ts_lshr_b32 s38, %1, 8\n\tv_mov_b32 v44, s38\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 30) {    // vector part = three vector instructions that read no SGPR
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, 5, v48\n\tv_and_b32 v42, 0xfff, v42\n\tv_add_u32 v44, 7, v48\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    } else if (MODE == 26) {    // the shipped step without the address mask (two vector instructions + ds_write_b8)
        unsigned long long buf = 0x0123456789abcdefull;
        unsigned e = 0x01004105u, pos = 0;
        asm volatile("v_lshlrev_b32 v40, 2, %3\n\t"
                     "ds_read_b32 v60, v40 offset:0\n\tds_read_b32 v61, v40 offset:256\n\tds_read_b32 v62, v40 offset:512\n\t"
                     "ds_read_b32 v63, v40 offset:768\n\tds_read_b32 v64, v40 offset:1024\n\tds_read_b32 v65, v40 offset:1280\n\t"
                     "ds_read_b32 v66, v40 offset:1536\n\tds_read_b32 v67, v40 offset:1792\n\tds_read_b32 v68, v40 offset:2048\n\t"
                     "ds_read_b32 v69, v40 offset:2304\n\tds_read_b32 v70, v40 offset:2560\n\tds_read_b32 v71, v40 offset:2816\n\t"
                     "ds_read_b32 v72, v40 offset:3072\n\tds_read_b32 v73, v40 offset:3328\n\tds_read_b32 v74, v40 offset:3584\n\t"
                     "ds_read_b32 v75, v40 offset:3840\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_lshlrev_b32 v48, 3, %3\n\tv_add_u32 v48, 8, v48\n\t"
                     "s_mov_b64 s[40:41], %0\n\ts_mov_b64 s[58:59], exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s39, 0x10000000\n\t"
                     "s_mov_b32 s42, 0x7fffffff\n\ts_mov_b32 s49, %4\n\t"
                     "s_set_gpr_idx_on s63, 0x1\n"
                     "2:\n\t"
                     "s_and_b32 s47, %1, 15\n\ts_lshr_b64 s[40:41], s[40:41], s47\n\ts_sub_u32 s42, s42, s47\n\ts_cmp_le_u32 s42, 32\n\t"
                     "s_cbranch_scc1 9f\n\t"
                     "s_bfe_u32 s60, s40, 0x40006\n\ts_and_b32 s61, s40, 63\n\ts_set_gpr_idx_idx s60\n\t"
                     "s_bfe_u32 s62, %1, 0x20018\n\ts_mov_b32 exec_lo, s62\n\tv_add_u32 v42, %2, %3\n\tv_bfe_u32 v44, %1, v48, 8\n\tds_write_b8 v42, v44 offset:4096\n\t"
                     "s_bcnt1_i32_b32 s47, s62\n\ts_add_u32 %2, %2, s47\n\t"
                     "v_readlane_b32 %1, v60, s61\n\ts_and_b32 %1, %1, 0x01ffffff\n\ts_or_b32 %1, %1, 0x01000005\n\t"
                     "s_or_b32 s40, s40, 0x300\n\ts_or_b32 s41, s41, 0x300\n\t"                 // keep the chase alive: the buffer never runs dry
                     "s_sub_u32 s49, s49, 1\n\ts_cmp_eq_u32 s49, 0\n\ts_cbranch_scc1 9f\n\t"
                     "s_cmp_lt_u32 %1, s39\n\ts_cbranch_scc1 2b\n"
                     "9:\n\t"
                     "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[58:59]"
                     : "+s"(buf), "+s"(e), "+s"(pos)
                     : "v"(threadIdx.x), "s"(iters)
                     : "s38", "s39", "s40", "s41", "s42", "s47", "s49", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v42", "v44", "v48", "v60", "v61", "v62",
                       "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "scc", "memory");
        idx = e + pos;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = idx;
}

template <int MODE>
static int run(const char *name, const unsigned *d_init, unsigned *d_out, int blocks, int iters, double per_iter_instr)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(64), 0, 0, d_init, 1000, d_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(64), 0, 0, d_init, iters, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned last = 0;
    CHECK(hipMemcpy(&last, d_out, sizeof last, hipMemcpyDeviceToHost));
    printf("%-36s %5d workgroups: %8.1f ns per iteration (%4.1f instructions in the chain; ends at %u)\n", name, blocks,
           ms * 1e6 / iters, per_iter_instr, last);
    return 0;
}

int main()
{
    std::vector<unsigned> init(kTab);
    for (int k = 0; k < kTab; ++k) init[k] = ((k * 37 + 11) % kTab) * 4 | 0x01000005u;   // a permutation, as byte offsets
    unsigned *d_init, *d_out;
    CHECK(hipMalloc(&d_init, sizeof(unsigned) * kTab));
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * 4096));
    CHECK(hipMemcpy(d_init, init.data(), sizeof(unsigned) * kTab, hipMemcpyHostToDevice));
    const int iters = 200000;
    for (int blocks : {1, 1024}) {          // one lone wavefront; four per CU on all 256 CUs (the decoder's occupancy)
        if (run<0>("lds_scalar", d_init, d_out, blocks, iters, 5)) return 1;
        if (run<1>("lds_scalar +8 salu", d_init, d_out, blocks, iters, 13)) return 1;
        if (run<2>("lds_scalar +3 taken branches", d_init, d_out, blocks, iters, 8)) return 1;
        if (run<3>("lds_vector", d_init, d_out, blocks, iters, 3)) return 1;
        if (run<4>("lds_vector +8 valu", d_init, d_out, blocks, iters, 11)) return 1;
        if (run<5>("readlane hop (+s_nop 4)", d_init, d_out, blocks, iters, 3)) return 1;
        if (run<6>("16 dependent salu", d_init, d_out, blocks, iters, 16)) return 1;
        if (run<7>("16 dependent valu", d_init, d_out, blocks, iters, 16)) return 1;
        if (run<8>("decoder literal step", d_init, d_out, blocks, iters, 14)) return 1;
        if (run<11>("decoder literal step, lgkmcnt(1)", d_init, d_out, blocks, iters, 14)) return 1;
        if (run<20>("literal step as shipped", d_init, d_out, blocks, iters, 24)) return 1;
        if (run<33>("  .. scalar first, one vector cluster", d_init, d_out, blocks, iters, 26)) return 1;
        if (run<21>("  .. without the exec write", d_init, d_out, blocks, iters, 23)) return 1;
        if (run<22>("  .. without the store", d_init, d_out, blocks, iters, 19)) return 1;
        if (run<23>("  .. without the lookup", d_init, d_out, blocks, iters, 20)) return 1;
        if (run<24>("  .. store = ds_write_b8 only", d_init, d_out, blocks, iters, 21)) return 1;
        if (run<25>("  .. store = 3 vector instr. only", d_init, d_out, blocks, iters, 23)) return 1;
        if (run<26>("  .. store without the mask", d_init, d_out, blocks, iters, 23)) return 1;
        if (run<27>("  .. vector part = v_add(pos) only", d_init, d_out, blocks, iters, 21)) return 1;
        if (run<28>("  .. vector part = v_bfe(entry) only", d_init, d_out, blocks, iters, 21)) return 1;
        if (run<29>("  .. vector part = s_lshr + v_mov(entry)", d_init, d_out, blocks, iters, 22)) return 1;
        if (run<30>("  .. vector part reads no SGPR", d_init, d_out, blocks, iters, 23)) return 1;
        if (run<31>("  .. no lookup, index mode never on", d_init, d_out, blocks, iters, 20)) return 1;
        if (run<32>("  .. index mode on only around lookup", d_init, d_out, blocks, iters, 27)) return 1;
        if (run<10>("lds_scalar, loop in asm", d_init, d_out, blocks, iters, 8)) return 1;
        if (run<9>("register table (gpr_idx + readlane)", d_init, d_out, blocks, iters, 10)) return 1;
    }
    return 0;
}
