// Host build of the wavefront DEFLATE decoder (nemoflux_amd/csrc/nf_inflate_core.h with NFI_HOST: one "lane", no barriers)
// for tests/test_inflate_cpu.py: the decoder's logic is checked against zlib's own streams on the CPU; the device build
// of the same source is checked on the GPU by tests/test_gpu_inflate.py.  Test infrastructure only.
#define NFI_HOST 1
#include <stdlib.h>
#include <string.h>

#include "../../nemoflux_amd/csrc/nf_inflate_core.h"

extern "C" int nfi_host_inflate(const unsigned char *src, unsigned in_len, unsigned readable, unsigned char *dst,
                                unsigned out_len)
{
    NfiCtx *c = (NfiCtx *)malloc(sizeof(NfiCtx));
    memset(c, 0, sizeof(NfiCtx));
    const int rc = nfi_inflate_stream(*c, src, in_len, readable, dst, out_len);
    free(c);
    return rc;
}
extern "C" unsigned nfi_host_ctx_bytes(void) { return (unsigned)sizeof(NfiCtx); }
