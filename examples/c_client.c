/* c_client.c -- a plain-C program on the C ABI of libnemoflux_amd.so (no Python, no torch): README.md:26-32, the
 * 36 x 18 x 1 x 1 grid with psi = x and the 6-point transect, whose flux is 360 (= psi(end) - psi(start)).
 *
 *   gcc -std=c99 -Iinclude examples/c_client.c -Lnemoflux_amd -lnemoflux_amd -Wl,-rpath,$PWD/nemoflux_amd -lm -o c_client
 *
 * Level 2 (nf_field_*): cell bounds + u, v on the host -> flux of the transect.
 * Level 1 (mnt_*): the same number through mint's call sequence on the (ncell,4) array Level 2 produced
 *                  (horizgrid.py:23-24, field.py:45-48,102).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "nemoflux_amd.h"

#define NX 36
#define NY 18
#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != NF_OK) {                                                           \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, nf_last_error());     \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static double psi(double x, double y) { (void)y; return x; }

/* great-circle length on the unit sphere (geo.py:14-27) */
static double arc(double lon0, double lat0, double lon1, double lat1)
{
    const double d2r = 3.14159265358979323846 / 180.0;
    double a[3] = {cos(lat0 * d2r) * cos(lon0 * d2r), cos(lat0 * d2r) * sin(lon0 * d2r), sin(lat0 * d2r)};
    double b[3] = {cos(lat1 * d2r) * cos(lon1 * d2r), cos(lat1 * d2r) * sin(lon1 * d2r), sin(lat1 * d2r)};
    return fabs(acos(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]));
}

int main(void)
{
    static double blon[NY][NX][4], blat[NY][NX][4], u[NY][NX], v[NY][NX], points[NY * NX][4][3];
    static double iV[NY * NX][4], eU[NY * NX], eV[NY * NX];
    const double dx = 360.0 / NX, dy = 180.0 / NY;
    for (int j = 0; j < NY; ++j)
        for (int i = 0; i < NX; ++i) {
            const double x0 = -180.0 + i * dx, x1 = x0 + dx, y0 = -90.0 + j * dy, y1 = y0 + dy;
            const double lon[4] = {x0, x1, x1, x0}, lat[4] = {y0, y0, y1, y1}; /* SW, SE, NE, NW (datagen.py:56-66) */
            for (int k = 0; k < 4; ++k) {
                blon[j][i][k] = lon[k];
                blat[j][i][k] = lat[k];
                points[j * NX + i][k][0] = lon[k];
                points[j * NX + i][k][1] = lat[k];
                points[j * NX + i][k][2] = 0.0;
            }
            /* datagen.py:92-113: u on the east edge, v on the north edge, from the stream function */
            double ds21 = arc(x1, y0, x1, y1), ds23 = arc(x1, y1, x0, y1);
            if (ds23 < 1e-12) ds23 = 1e-12;
            u[j][i] = (psi(x1, y1) - psi(x1, y0)) / ds21;
            v[j][i] = -(psi(x1, y1) - psi(x0, y1)) / ds23;
        }
    const double thickness[1] = {1.0};
    const double xyz[6][3] = {{-180, -70, 0}, {-160, -10, 0}, {-35, 40, 0}, {20, -50, 0}, {60, 50, 0}, {180, 40, 0}};

    char name[128];
    CHECK(nf_device_name(name, (int)sizeof name));
    printf("device: %s, library version %d\n", name, nf_version());

    /* ---- Level 2 */
    nf_field *fld = NULL;
    int tid = -1, rowlen = 0, nseg = 0;
    CHECK(nf_field_new(&fld));
    CHECK(nf_field_set_bounds(&fld, blon, blat, NY, NX, NF_F64, 0));
    CHECK(nf_field_set_thickness(&fld, thickness, 1));
    CHECK(nf_field_set_uv(&fld, u, v, 1, NF_F64, 0, NAN));
    CHECK(nf_field_add_transect(&fld, &xyz[0][0], 6, 0, &tid));
    CHECK(nf_field_build_weights(&fld, 128, 360.0));
    CHECK(nf_field_row_length(&fld, &rowlen));
    CHECK(nf_field_num_segments(&fld, &nseg));
    double *row = (double *)calloc((size_t)rowlen, sizeof(double));
    CHECK(nf_field_compute_flux(&fld, 0, row));
    double maxabs = 0.0;
    CHECK(nf_field_read_step(&fld, &iV[0][0], eU, eV, &maxabs));
    const double flux2 = row[nseg + tid];
    printf("level 2: %d segments, flux = %.12f (max |edge flux| %.6g)\n", nseg, flux2, maxabs);

    /* ---- Level 1: mint's call sequence on the host array */
    Grid_t *grid = NULL;
    PolylineIntegral_t *pli = NULL;
    size_t ncells = 0;
    double flux1 = 0.0;
    CHECK(mnt_grid_new(&grid));
    CHECK(mnt_grid_setPointsPtr(&grid, &points[0][0][0]));
    CHECK(mnt_grid_build(&grid, 4, (long long)NY * NX));
    CHECK(mnt_grid_getNumberOfCells(&grid, &ncells));
    CHECK(mnt_polylineintegral_new(&pli));
    CHECK(mnt_polylineintegral_setGrid(&pli, grid));
    CHECK(mnt_polylineintegral_buildLocator(&pli, 128, 360.0, 0));
    CHECK(mnt_polylineintegral_computeWeights(&pli, 6, &xyz[0][0], 0));
    CHECK(mnt_polylineintegral_getIntegral(&pli, &iV[0][0], MNT_CELL_BY_CELL_DATA, &flux1));
    printf("level 1: %zu cells, flux = %.12f\n", ncells, flux1);

    /* error convention: non-zero return + message, nothing thrown */
    const int rc = mnt_polylineintegral_computeWeights(&pli, 1, &xyz[0][0], 0);
    printf("computeWeights with 1 point -> %d (%s)\n", rc, nf_last_error());

    /* ---- file ingest: one zlib stream (a stored block holding the float 1.0f, Adler-32 0x014300C0) inflated on the device */
    static const unsigned char stream[15] = {0x78, 0x01, 0x01, 0x04, 0x00, 0xFB, 0xFF, 0x00, 0x00, 0x80, 0x3F, 0x01, 0x43, 0x00, 0xC0};
    const long long in_off[1] = {0}, in_len[1] = {15}, dims[3] = {1, 1, 1}, origin[3] = {0, 0, 0};
    nf_inflater *inf = NULL;
    void *slab = NULL;
    float decoded = 0.f;
    int status = -1;
    CHECK(nf_inflater_new(&inf));
    CHECK(nf_malloc(&slab, 16));
    CHECK(nf_inflater_run(&inf, stream, sizeof stream, in_off, in_len, 1, 4, 4, 0, dims, dims, origin, slab, NULL, &status));
    CHECK(nf_memcpy_d2h(&decoded, slab, 4));
    CHECK(nf_free(slab));
    CHECK(nf_inflater_del(&inf));
    printf("ingest: decoded %.1f (status %d)\n", decoded, status);

    /* ---- the multi-GPU collective from plain C: a one-rank communicator (all one GPU can form), the row summed in place.
     * With N ranks: every rank runs the non-collective preflight, the ranks agree that all of them passed (MPI_Allreduce
     * MIN, say), rank 0 makes the id and hands it to the others (MPI, a file, a socket), and only then does every rank call
     * comm_init -- ncclCommInitRank is collective, a rank that failed earlier would leave the others waiting inside it. */
    char id[NF_RCCL_UNIQUE_ID_BYTES];
    void *comm = NULL, *row_dev = NULL;
    int nranks = -1, myrank = -1, dev = -1;
    double reduced[3] = {0, 0, 0};
    const double part[3] = {1.5, -2.0, 360.0};
    int pre_dev = -1;
    CHECK(nf_rccl_preflight(&pre_dev));
    CHECK(nf_rccl_unique_id(id));
    CHECK(nf_rccl_comm_init(&comm, 1, id, 0));
    CHECK(nf_rccl_comm_info(comm, &nranks, &myrank, &dev));
    CHECK(nf_malloc(&row_dev, sizeof part));
    CHECK(nf_memcpy_h2d(row_dev, part, sizeof part));
    CHECK(nf_rows_allreduce(comm, (double *)row_dev, 3, NULL));
    CHECK(nf_synchronize());
    CHECK(nf_memcpy_d2h(reduced, row_dev, sizeof reduced));
    CHECK(nf_free(row_dev));
    CHECK(nf_rccl_comm_destroy(comm));
    printf("reduce: %d rank(s), rank %d on device %d, row = %.1f %.1f %.1f\n", nranks, myrank, dev, reduced[0], reduced[1], reduced[2]);

    CHECK(mnt_polylineintegral_del(&pli));
    CHECK(mnt_grid_del(&grid));
    CHECK(nf_field_del(&fld));
    free(row);
    const int ok = fabs(flux2 - 360.0) < 1e-9 && fabs(flux1 - 360.0) < 1e-9 && rc == NF_ERR_ARG && ncells == NY * NX &&
                   decoded == 1.0f && status == 0 && nranks == 1 && myrank == 0 && reduced[0] == 1.5 && reduced[2] == 360.0;
    printf(ok ? "C client OK\n" : "C client FAILED\n");
    return ok ? 0 : 2;
}
