/*
 * mpassit_amd.h -- C-ABI of the MI355X-native regrid engine that replaces the ESMF calls made by
 * LarissaReames-NOAA/MPASSIT's hot path (interp.F90 + the ESMF_Mesh/Grid objects of model_grid.F90).
 *
 * Every entry point replaces one ESMF verb at the cited reference call sites.
 * Conventions (mirroring the reference, utils.F90:16-33 / interp.F90:130-131):
 *   - every function returns an int rc, 0 == MPG_SUCCESS (like ESMF_SUCCESS); on failure
 *     mpg_last_error() returns a message; the caller decides to abort (the reference always does,
 *     error_handler -> mpi_abort(999)).
 *   - the caller owns all host buffers; the library owns device buffers and the opaque objects.
 *   - one process drives one GPU (one "PET" per GPU, mpassit.F90:84-96); calls are synchronous at the
 *     boundary unless the name ends in _dev and a stream is passed.
 *   - all floating point data is float64 (ESMF_TYPEKIND_R8 everywhere, interp.F90:479), indices int32.
 *   - there is NO CPU fallback: without a HIP device mpg_init fails and every other call returns
 *     MPG_ERR_NOT_INITIALIZED.
 */
#ifndef MPASSIT_AMD_H
#define MPASSIT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpg_mesh_s *mpg_mesh;     /* replaces type(ESMF_Mesh)        model_grid.F90:72  */
typedef struct mpg_grid_s *mpg_grid;     /* replaces type(ESMF_Grid)        model_grid.F90:73  */
typedef struct mpg_handle_s *mpg_handle; /* replaces type(ESMF_RouteHandle) interp.F90:86,192  */

enum {
  MPG_SUCCESS = 0,
  MPG_ERR_NOT_INITIALIZED = 1,
  MPG_ERR_INVALID_ARG = 2,
  MPG_ERR_HIP = 3,
  MPG_ERR_UNSUPPORTED = 4,
  MPG_ERR_OVERFLOW = 5,
  MPG_ERR_TIMEOUT = 6   /* a wait on another rank hit its deadline (mpg_comm_*): exit the process, do not retry in it */
};

/* ESMF_REGRIDMETHOD_* (interp.F90:119,204,370,420) */
enum { MPG_REGRIDMETHOD_BILINEAR = 0, MPG_REGRIDMETHOD_CONSERVE = 1, MPG_REGRIDMETHOD_NEAREST_STOD = 2 };
/* ESMF_MESHLOC_* (input_data.F90:927,1116-1123) */
enum { MPG_MESHLOC_ELEMENT = 0, MPG_MESHLOC_NODE = 1 };
/* ESMF_STAGGERLOC_* (interp.F90:480,489,509; model_grid.F90:706-728) */
enum { MPG_STAGGERLOC_CENTER = 0, MPG_STAGGERLOC_EDGE1 = 1, MPG_STAGGERLOC_EDGE2 = 2, MPG_STAGGERLOC_CORNER = 3 };
/* memory order of a source field with an ungridded (level) dimension */
enum {
  MPG_LAYOUT_CELL_FAST = 0, /* [nlev][ncell]: how the reference holds fields, input_data.F90:653-655 */
  MPG_LAYOUT_LEV_FAST = 1   /* [ncell][nlev]: MPAS file order, input_data.F90:630,645 (fused ingest) */
};

/* ---- runtime: ESMF_Initialize / ESMF_Finalize (mpassit.F90:84,140) ------------------------------ */
/* mpg_init may be repeated with the SAME device index (no-op); a different index while initialised is refused with
 * MPG_ERR_INVALID_ARG (streams and pinned staging belong to the first device): mpg_finalize first.
 * MPASSIT is a single-shot tool (mpassit.F90:105-137), so what a run pays are FIRST calls: mpg_init starts a helper thread that
 * loads the library's code objects and warms the runtime's pageable-copy staging while the caller goes on reading its
 * namelist and opening its files (the HIP runtime would otherwise load each translation unit at the first launch of one of
 * its kernels: 5-10 ms in front of the first Store of each method).  MPG_NO_WARMUP=1 in the environment switches it off. */
int mpg_init(int device);
/* Blocks until that helper thread is done (a few ms to a few hundred after mpg_init).  The thread allocates, frees and copies while it
 * runs, which a stream capture in hipStreamCaptureModeGlobal does not tolerate from another thread: a host that starts capturing
 * right after mpg_init calls this first (mpassit_amd/interp.py GraphedInterp does).  No-op when there is no such thread. */
int mpg_warmup_wait(void);
int mpg_finalize(void);
const char *mpg_last_error(void);
/* number of GPUs the process sees (0 when there is none); needs no mpg_init: a launcher's ranks choose their device with it */
int mpg_device_count(int *n);
/* "gfx950" etc.; buf may be NULL */
int mpg_device_info(char *arch_buf, int buf_len, int *n_cu, int64_t *hbm_bytes);

/* ---- ESMF_MeshCreate (model_grid.F90:488-497) ---------------------------------------------------
 * Arrays exactly as read from the MPAS file (model_grid.F90:354-417): lat/lon in RADIANS, the
 * deg conversion + (-180,180] wrap of :450-454,464-468 happens inside; verticesOnCell is
 * [nCells][maxEdges], 1-based, 0-padded (:448,479).  Elements = cells, nodes = vertices.
 * Refused with MPG_ERR_INVALID_ARG (and the offending index in mpg_last_error) before any geometry kernel runs: a vertex number
 * beyond nVertices, a coordinate that is NaN / Inf, a latitude beyond +-pi/2 (degrees handed over as radians). */
int mpg_mesh_create(int64_t nCells, int64_t nVertices, int maxEdges, const double *latCell,
                    const double *lonCell, const double *latVertex, const double *lonVertex,
                    const int32_t *verticesOnCell, mpg_mesh *out);
int mpg_mesh_destroy(mpg_mesh mesh); /* ESMF_MeshDestroy model_grid.F90:2154 */
/* The same for ONE rank of a job whose target rows are sharded over several GPUs: `grid` is this rank's row block (with the
 * halo rows it regrids itself), and only the part of the mesh that grid can see is brought to the device -- where the
 * reference hands every rank 1/N of the cells (para_range, model_grid.F90:423-438, 2428-2441) and ESMF redistributes.  The
 * host passes the same whole arrays; all cell CENTRES are uploaded (16 B per cell) and classified against the grid on the
 * device, then verticesOnCell, the vertex coordinates, the dual triangles and the nearest-neighbour BVH exist for the
 * covering id range of the cells within a margin of the grid only (spatially banded numbering makes that range tight;
 * arbitrary numbering degrades towards the whole mesh, never towards a wrong answer).  Ids stay GLOBAL: handles, source
 * ranges, windows and halo schedules are those of mpg_mesh_create, and every RegridStore of this mesh onto `grid` gives
 * bit-identical weights -- the window's closure is verified on the device and widened until it holds
 * (csrc/k_mesh_window.hip).  Stores onto any other grid are refused; `grid` must outlive the mesh's Stores.
 * mpg_mesh_window_info: the cell rows [cell_first, +cell_count) and vertices [vertex_first, +vertex_count) that are
 * resident, and the chord distance `margin` from the grid within which every cell is (any pointer may be NULL). */
int mpg_mesh_create_window(int64_t nCells, int64_t nVertices, int maxEdges, const double *latCell, const double *lonCell,
                           const double *latVertex, const double *lonVertex, const int32_t *verticesOnCell, mpg_grid grid, mpg_mesh *out);
int mpg_mesh_window_info(mpg_mesh mesh, int64_t *cell_first, int64_t *cell_count, int64_t *vertex_first, int64_t *vertex_count, double *margin);

/* ---- ESMF_GridCreateNoPeriDim / 1PeriDim + GridAddCoord x4 (model_grid.F90:684-728,736-1038) ------
 * nx, ny = mass (CENTER) point counts (i_target, j_target).  Coordinates in DEGREES, C order with i
 * fastest: centre [ny][nx], corner [ny+1][nx+1], EDGE1 (U) [ny][nx+1], EDGE2 (V) [ny+1][nx].
 * corner/edge arrays may be NULL when the corresponding stagger is never used.
 * periodic_i = 0: ESMF_GridCreateNoPeriDim (regional, model_grid.F90:699).
 * periodic_i = MPG_GRID_PERIODIC_I (1): ESMF_GridCreate1PeriDim(periodicDim=1, poleDim=2, MONOPOLE at both
 *   ends, model_grid.F90:685-694): CENTER column nx-1 neighbours column 0, and each j end is closed by a pole
 *   node at lat -/+90 whose value is the mean of the first / last CENTER row.  Array shapes are the same as
 *   in the regional case; the extra EDGE1 / CORNER column (index nx) duplicates column 0 one period later
 *   (ESMF's periodic staggers hold only the first nx columns).  Only Grid -> Grid RegridStore reads the flag.
 * OR in MPG_GRID_NO_SOUTH_POLE / MPG_GRID_NO_NORTH_POLE for a row block of a periodic grid that does not
 *   touch that pole (multi-GPU row sharding).
 * A coordinate that is NaN / Inf (or |lat| > 180) is refused with MPG_ERR_INVALID_ARG and the point's index; latitudes a little beyond
 * the pole (the corner row of a global lat-lon grid) are angles and pass. */
enum { MPG_GRID_PERIODIC_I = 1, MPG_GRID_NO_SOUTH_POLE = 2, MPG_GRID_NO_NORTH_POLE = 4 };
int mpg_grid_create(int nx, int ny, int periodic_i, const double *lon_center, const double *lat_center,
                    const double *lon_corner, const double *lat_corner, const double *lon_edge1,
                    const double *lat_edge1, const double *lon_edge2, const double *lat_edge2,
                    mpg_grid *out);
int mpg_grid_destroy(mpg_grid grid); /* ESMF_GridDestroy model_grid.F90:2156 */

/* ---- target grid straight from the projection, on the device -------------------------
 * Replaces the host loops of define_target_grid_params (model_grid.F90:736-1038): get_lat_lon_fields x4 staggers
 * (:2188-2219 -> xytoll, llxy_module.F90:166-216 -> ij_to_latlon, module_map_utils.F90:629-679, Lambert :1160-1233,
 * lat-lon :1398-1428), get_rotang (:2450-2507) and get_map_factor (:2229-2365).  The fields of mpg_proj are the
 * arguments of push_source_projection / map_set (model_grid.F90:676-678); the derived constants (cone, rsw, pole
 * i/j; set_lc, module_map_utils.F90:1083-1121) are computed inside.  nx, ny = mass point counts (i_target,
 * j_target); stagger shapes as in mpg_grid_create.  The grid keeps lon/lat (degrees), cos/sin(alpha) and the map
 * factors on the device; the getters copy them to the host (XLAT/XLONG/MAPFAC/SINALPHA/COSALPHA of the output file,
 * write_data.F90:1003-1140).
 * PROJ_PS (polar stereographic: truelat1, stand_lon, dx_m; set_ps / ijll_ps, module_map_utils.F90:682-822) and PROJ_MERC
 * (Mercator: truelat1, dx_m; set_merc / ijll_merc, :1293-1362) take the arguments push_source_projection passes for them
 * (llxy_module.F90:71-79,123-132).
 * Map factors: PROJ_LC / PROJ_PS / PROJ_MERC as get_map_factor; PROJ_LATLON has no branch there (the reference writes unset memory) and
 * returns 1.0 here.  cos/sin(alpha) exist for PROJ_LC only (model_grid.F90:1113), as in the reference. */
enum { MPG_PROJ_LATLON = 0, MPG_PROJ_LC = 1, MPG_PROJ_PS = 2, MPG_PROJ_MERC = 3 }; /* misc_definitions_module.F90:38-42 */
typedef struct mpg_proj {
  int code;
  double known_lat, known_lon, known_x, known_y; /* lat1, lon1, knowni, knownj */
  double dx_m;                                   /* PROJ_LC, PROJ_PS, PROJ_MERC: grid spacing in metres */
  double stand_lon, truelat1, truelat2;          /* PROJ_LC; PROJ_PS: stand_lon, truelat1; PROJ_MERC: truelat1 */
  double dlat_deg, dlon_deg;                     /* PROJ_LATLON: latinc, loninc */
} mpg_proj;
int mpg_grid_create_proj(const mpg_proj *proj, int nx, int ny, int periodic_i, mpg_grid *out);
/* A grid made from coordinate ARRAYS (mpg_grid_create) whose arrays are rows row0 .. row0 + ny - 1 of the grid `proj` describes
 * (a rank's block of target rows, model_grid.F90:693): the Stores of Mesh -> Grid handles then find a source triangle's /
 * cell's target points through the inverse projection in O(1) instead of descending the box pyramid (PROJ_LC and PROJ_LATLON;
 * other projections are accepted and change nothing).  Every candidate is still tested exactly as before: the weights are
 * those of the pyramid search.  The claim is checked -- a sample of the grid's own CENTER points must fall on their own
 * indices -- and a projection that does not fit is refused with MPG_ERR_INVALID_ARG.  Grids of mpg_grid_create_proj have
 * their inverse from the start. */
int mpg_grid_attach_proj(mpg_grid grid, const mpg_proj *proj, int row0);
int mpg_grid_get_coords(mpg_grid grid, int staggerloc, double *lon_host, double *lat_host);
int mpg_grid_get_rotang(mpg_grid grid, double *cosa_host, double *sina_host);
int mpg_grid_get_mapfac(mpg_grid grid, int staggerloc, double *mapfac_host);
/* device pointers owned by the grid ([ny][nx], CENTER), for mpg_rotate_winds_dev */
int mpg_grid_rotang_dev(mpg_grid grid, const double **cosa_dev, const double **sina_dev);

/* ---- ESMF_Field[Bundle]RegridStore (interp.F90:123,207,226,241,259,277,334,353,372,394,421,437) ---
 * srcTermProcessing=1, unmappedaction=IGNORE are implied.  Mesh -> Grid.  Handles are cached: the same
 * (mesh, src_loc, grid, dst_stagger, method) returns the same handle (reference recomputes it up to
 * 13x per run); each Store must be paired with one mpg_handle_release.  A handle whose last reference
 * was released keeps its weights in the cache (up to 8 such handles; dropped when their mesh or grid is destroyed or
 * at mpg_finalize), so the Store / Regrid / Release / Store-again sequence of interp_diag_data followed by
 * interp_hist_data (interp.F90:123-148, :207) builds the element -> CENTER bilinear weights once. */
int mpg_regrid_store(mpg_mesh src, int src_meshloc, mpg_grid dst, int dst_staggerloc, int regridmethod,
                     mpg_handle *out);
/* Grid -> Grid on the same grid, CENTER -> EDGE1/EDGE2 bilinear (interp.F90:298,316). */
int mpg_regrid_store_grid(mpg_grid grid, int src_staggerloc, int dst_staggerloc, int regridmethod,
                          mpg_handle *out);
/* The same two Stores STARTED and not waited for.  interp.F90:207-437 stores its weight sets one after the other, each in front of
 * the Regrids that use it; they are independent of each other and of every Regrid that does not use them.  A _begin call queues
 * the Store on the library's worker thread (own stream, one Store at a time) and returns; the matching mpg_regrid_store[_grid]
 * -- same arguments -- later returns the finished handle, waiting only for what is left of it.  A host that begins its
 * conservative, nearest-neighbour and destaggering Stores before it regrids its bilinear fields hides them behind those Regrids
 * (bench.py `job` leg).  Weights, cache and reference counting are those of the plain calls; a Store that nobody collects stays
 * a parked cache entry.  Errors of a begun Store surface in the collecting call. */
int mpg_regrid_store_begin(mpg_mesh src, int src_meshloc, mpg_grid dst, int dst_staggerloc, int regridmethod);
int mpg_regrid_store_grid_begin(mpg_grid grid, int src_staggerloc, int dst_staggerloc, int regridmethod);

/* ---- ESMF_Field[Bundle]Regrid (interp.F90:134,219,236,251,268,286,307,325,344,363,382,404,431,443) --
 * dst is fully overwritten: [nfields][nlev][ny_dst][nx_dst], unmapped points = 0.0 (zeroregion=TOTAL).
 * src: nfields slabs of nlev*n_src doubles in `src_layout` order.  Host-pointer version copies
 * H2D/D2H internally; the _dev version takes device pointers and a hipStream_t (NULL = default
 * stream) and returns after enqueueing.  The first _dev call on a handle for a given layout decides the kernel and
 * may build its tile lists (allocation + synchronisation); every later call only enqueues kernels on the stream, so a
 * caller can capture its per-time-level sequence of _dev calls (Regrid, rotate_winds, post-ops) into a hipGraph. */
int mpg_regrid(mpg_handle rh, const double *src_host, int src_layout, int nlev, int nfields, double *dst_host);
int mpg_regrid_dev(mpg_handle rh, const double *src_dev, int src_layout, int nlev, int nfields,
                   double *dst_dev, void *hip_stream);
/* Fused ingest/egress Regrid (the callers either side of the hot path): the source may be
 * float32 (MPAS history variables are single precision; the reference widens them at read, input_data.F90:630-655)
 * and the destination float32 (every output variable is NF90_FLOAT, write_data.F90:779).  The arithmetic stays
 * float64 and  dst = (dst type)( regrid(src) * scale + offset )  reproduces the writer's post-ops (T - 300,
 * write_data.F90:1343; PHB * 9.81, :1418), so float32 results are bit-identical to the reference's file contents.
 * src_type / dst_type: MPG_TYPE_F64 or MPG_TYPE_F32, optionally | MPG_TYPE_BE: the values are big-endian in memory, as a
 * NetCDF classic (CDF-1/2/5) variable stores them -- the bytes of nf90_get_var's source (input_data.F90:630) and of
 * nf90_put_var's destination (write_data.F90:1339-1475) travel file <-> HBM untouched (mpg_file_to_dev / mpg_dev_to_file)
 * and the Regrid reads / writes them as they are: no byte-swap pass over the data exists.
 * Device pointers, stream as in mpg_regrid_dev. */
enum { MPG_TYPE_F64 = 0, MPG_TYPE_F32 = 1, MPG_TYPE_BE = 2 };
int mpg_regrid_typed_dev(mpg_handle rh, const void *src_dev, int src_type, int src_layout, int nlev, int nfields,
                         void *dst_dev, int dst_type, double scale, double offset, void *hip_stream);
/* ESMF_FieldBundleRegrid as interp.F90:240-254 issues it: ONE Regrid over every field of a bundle whose fields are SEPARATE
 * arrays (an ESMF bundle holds independent fields; here: nfields device pointers on either side, host arrays of pointers).
 * All fields share the handle, the layout, nlev and the element types; offsets (nfields values, or NULL for 0) is the
 * epilogue offset per field (T - 300 beside fields written as they are).  The same kernels as mpg_regrid_typed_dev with
 * nfields consecutive slabs -- one launch, the per-point indices and weights shared by the fields from the L2 -- and the same
 * bits as nfields single calls; 10 % (configuration 4) to 18 % (configuration 5) faster than those. */
int mpg_regrid_bundle_typed_dev(mpg_handle rh, int nfields, const void *const *src_dev, int src_type, int src_layout, int nlev,
                                void *const *dst_dev, int dst_type, double scale, const double *offsets, void *hip_stream);
/* The same on HOST buffers (pageable memory: Fortran allocatables, numpy arrays), for hosts that keep the reference's
 * file -> host array -> regrid -> host array -> file shape and are therefore bound by the PCIe link: float32 sources
 * and results cross the link as they are stored in the files (half the bytes of the float64 route), and the field is cut
 * into chunks whose upload, kernel and download overlap (full duplex, a helper thread downloads while the caller
 * uploads).  Blocks until dst_host is complete. */
int mpg_regrid_typed(mpg_handle rh, const void *src_host, int src_type, int src_layout, int nlev, int nfields,
                     void *dst_host, int dst_type, double scale, double offset);
/* ... and over a bundle whose fields are SEPARATE host arrays (mpg_regrid_bundle_typed_dev's host twin): all fields go through
 * one pipeline -- the upload of field k + 1, the Regrid of field k and the download of field k - 1 overlap -- where a file-order
 * field handed over alone is uploaded, regridded and downloaded one step after the other (configuration 4, float64: 39 ms per
 * field alone, 24 in a bundle).  offsets: one epilogue offset per field, or NULL.  Blocks until every destination is complete. */
int mpg_regrid_bundle_typed(mpg_handle rh, int nfields, const void *const *src_host, int src_type, int src_layout, int nlev,
                            void *const *dst_host, int dst_type, double scale, const double *offsets);
/* ESMF_FieldBundleRegridRelease (interp.F90:450,455,461) */
int mpg_handle_release(mpg_handle rh);

/* ---- rotate_winds_cgrid (interp.F90:689-749): in place on CENTER-stagger u, v [nlev][ny][nx] with
 * cosalpha/sinalpha [ny][nx] (model_grid.F90:1154).  */
int mpg_rotate_winds(int64_t npts, int nlev, const double *cosa_host, const double *sina_host,
                     double *u_host, double *v_host);
int mpg_rotate_winds_dev(int64_t npts, int nlev, const double *cosa_dev, const double *sina_dev,
                         double *u_dev, double *v_dev, void *hip_stream);
/* The whole wind chain of interp_hist_data in ONE pass over the mass-point winds (interp.F90:291-328): rotate_winds_cgrid on
 * (UMASS, VMASS) -- :291-293, 737-748 -- followed by the two Grid -> Grid Regrids UMASS(CENTER) -> U(EDGE1) -- :295-311 -- and
 * VMASS(CENTER) -> V(EDGE2) -- :313-328.  u/v_target_grid_nostag are intermediates the reference never writes, so the earth-relative
 * mass winds are read once and only U and V are stored; the results are bit-identical to mpg_rotate_winds_dev followed by
 * mpg_regrid_dev (dst_type MPG_TYPE_F64) or mpg_regrid_typed_dev(..., dst_type, 1.0, 0.0) (any other dst_type) on the two handles.
 *   rh_edge1, rh_edge2   handles of mpg_regrid_store_grid(grid, CENTER, EDGE1 / EDGE2) on ONE grid; either may be NULL (that
 *                        component is not produced: do_u_interp / do_v_interp, interp.F90:295,313)
 *   cosa_dev, sina_dev   [ny][nx] as for mpg_rotate_winds_dev, or both NULL: no rotation (proj_code /= PROJ_LC, or one component)
 *   umass_dev, vmass_dev [nlev][ny][nx] float64, NOT modified
 *   u_dev, v_dev         [nlev][ny][nx+1], [nlev][ny+1][nx] of dst_type
 *   umass_rot_dev, vmass_rot_dev   optional (NULL): the rotated mass winds [nlev][ny][nx] as mpg_rotate_winds_dev would have left
 *                        them in place; must not be the input arrays
 * MPG_ERR_UNSUPPORTED: the handles are not such a pair (re-indexed by mpg_handle_localize / rebase, different grids): use the
 * three separate calls. */
int mpg_wind_destagger_dev(mpg_handle rh_edge1, mpg_handle rh_edge2, const double *cosa_dev, const double *sina_dev,
                           const double *umass_dev, const double *vmass_dev, int nlev, void *u_dev, void *v_dev, int dst_type,
                           double *umass_rot_dev, double *vmass_rot_dev, void *hip_stream);
/* The same chain for a host that keeps its fields in HOST arrays, as the reference does (farrayPtr in, farrayPtr out: interp.F90:702-735
 * fetches the pointers, :291-328 runs the chain).  The three separate host calls -- mpg_rotate_winds, then mpg_regrid on each handle --
 * move 4 fields up and 4 down over the link; this one moves the earth-relative mass winds up ONCE in chunks of levels and only U and V
 * down, both directions at once (2 up, 2 down: the chain is link-bound, so about half the time).  Same bits as mpg_wind_destagger_dev.
 * All pointers are host pointers; cosa_host / sina_host [ny][nx] or both NULL; umass_rot_host / vmass_rot_host optional (NULL) and --
 * unlike the device form -- MAY be the input arrays themselves: rotate_winds_cgrid's in-place result (a level's rotated winds come
 * down after that level went up).  Blocks until every destination is complete. */
int mpg_wind_destagger(mpg_handle rh_edge1, mpg_handle rh_edge2, const double *cosa_host, const double *sina_host,
                       const double *umass_host, const double *vmass_host, int nlev, void *u_host, void *v_host, int dst_type,
                       double *umass_rot_host, double *vmass_rot_host);

/* ---- output epilogues: what write_data.F90 computes on rank 0 between ESMF_FieldGather and nf90_put_var, done on
 * the device-resident regridded fields so that only final float32 arrays leave the GPU.
 * Every output variable is NF90_FLOAT (write_data.F90:312-980) while the fields are float64: the cast below is the
 * conversion nf90_put_var applies (round to nearest).  Device pointers; hip_stream as in mpg_regrid_dev.  dst_be != 0
 * stores the float32 results big-endian (the variable's bytes in a NetCDF classic file), as MPG_TYPE_BE does for Regrid.
 *   mpg_post_cast_dev        dst = (float)(src*scale + offset): plain fields (scale 1, offset 0), T - 300 (:1339-1347,
 *                            the `continue` in that loop is a no-op statement, so every point is shifted), PHB*9.81 (:1418)
 *   mpg_post_layer_mean_dev  Z_C(k) = 0.5*(PHB(k+1) + PHB(k)), src [nlevp1][n_pts] -> dst [nlevp1-1][n_pts] (:1406-1415)
 *   mpg_post_ptop_dev        P_TOP from P_HYD [nlev][n_pts]: min(maxval(P_HYD), 0.8*P_HYD(top) over columns whose top
 *                            value is >= 10) (:1362-1371); float64 result returned to the host, blocks on the stream */
/* ---- device buffers for hosts without a HIP binding of their own ---------------------------------------------------
 * ESMF owned the field storage (ESMF_FieldCreate, farrayPtr: input_data.F90:638, interp.F90:702).  A host that keeps its
 * fields in HBM between the input and the output file (fortran/mpassit_driver.F90 on NetCDF files) allocates them
 * here and passes the pointers to the _dev entry points.  upload / download are plain blocking copies. */
int mpg_dev_alloc(int64_t nbytes, void **out_dev);
int mpg_dev_free(void *dev);
int mpg_dev_upload(void *dst_dev, const void *src_host, int64_t nbytes);
int mpg_dev_download(void *dst_host, const void *src_dev, int64_t nbytes);

/* In-place byte swap of n elements of elem_size 2, 4 or 8 bytes on the device, for big-endian file data that does NOT
 * pass through mpg_regrid_typed_dev / mpg_post_*_dev (those take and produce it directly, MPG_TYPE_BE). */
int mpg_bswap_dev(void *buf_dev, int64_t n, int elem_size, void *hip_stream);
/* The transport of such a variable: bytes [offset, offset + nbytes) of a file -> device memory and back, untouched,
 * through pinned staging buffers and a few pread / pwrite threads (the page-cache side of the copy is what limits a
 * single core: eight readers -- MPG_IO_READ_THREADS=1..8 in the environment overrides --, four writers).  Replaces the nf90_get_var / nf90_put_var data movement of input_data.F90:630 and
 * write_data.F90:1008-1475 for NetCDF classic files; offset / nbytes come from ncio_var_extent.  Blocking: hip_stream is
 * synchronised first (earlier users / the producer of the device buffer), the range is complete on return.  The file
 * must already have the size (ncio_var_extent extends a file being written).  One read and one write may run
 * concurrently from two host threads. */
int mpg_file_to_dev(const char *path, int64_t offset, int64_t nbytes, void *dst_dev, void *hip_stream);
int mpg_dev_to_file(const char *path, int64_t offset, int64_t nbytes, const void *src_dev, void *hip_stream);
int mpg_post_cast_dev(const double *src_dev, int64_t n, double scale, double offset, float *dst_dev, int dst_be, void *hip_stream);
int mpg_post_layer_mean_dev(const double *src_dev, int nlevp1, int64_t n_pts, float *dst_dev, int dst_be, void *hip_stream);
int mpg_post_ptop_dev(const double *p_hyd_dev, int nlev, int64_t n_pts, double *ptop_host, void *hip_stream);
/* The two reductions P_TOP is made of, for a host that holds only a block of the grid rows (one driver image per GPU):
 * vmax = maxval(P_HYD) of the block, candmin = min of 0.8*P_HYD(top) over its columns with P_HYD(top) >= 10 (has_cand = 0
 * when there is none).  P_TOP = min(max of the vmax, min of the candmin) over the blocks -- exact, independent of the split. */
int mpg_post_ptop_parts_dev(const double *p_hyd_dev, int nlev, int64_t n_pts, double *vmax_host, double *candmin_host, int *has_cand_host,
                            void *hip_stream);

/* ---- bring-your-own weights: the factorList / factorIndexList form of ESMF_FieldRegridStore (and of an
 * ESMF_RegridWeightGen file: S, col, row).  Builds a route handle that applies externally computed weights with the
 * same Regrid kernels, e.g. to compare this library's weight generation with ESMF's on a site that has ESMF.
 * col = source index, row = destination index (j*nx_dst + i), both 1-based as in ESMF; any entry order (order
 * inside a destination row is kept = summation order).  Destination points without entries regrid to 0.0. */
int mpg_handle_from_weights(int64_t n_src, int nx_dst, int ny_dst, int64_t nnz, const int32_t *row_host,
                            const int32_t *col_host, const double *S_host, mpg_handle *out);

/* ---- introspection (tests, INTEGRATION.md, multi-GPU halo schedule) -------------------------------- */
/* n_src: source points the handle indexes; n_dst = nx_dst*ny_dst; nnz_per_row: 3 bilinear(mesh),
 * 4 bilinear(grid), 1 nearest, 0 = CSR (conservative); nnz = total stored weights. */
int mpg_handle_info(mpg_handle rh, int64_t *n_src, int64_t *n_dst, int *nx_dst, int *ny_dst,
                    int *nnz_per_row, int64_t *nnz);
/* fixed-nnz handles: idx/w are [n_dst][nnz_per_row] (row-major, host); idx = -1 where unmapped */
int mpg_handle_get_weights(mpg_handle rh, int32_t *idx_host, double *w_host);
/* CSR handles: rowptr [n_dst+1], col/val [nnz] (host) */
int mpg_handle_get_csr(mpg_handle rh, int64_t *rowptr_host, int32_t *col_host, double *val_host);
/* Which Regrid kernel serves a 3-point (bilinear) handle, decided when its tile lists are built on the first Regrid:
 * cell_fast_kernel / lev_fast_kernel: 0 = not decided yet (no bundle of >= 8 levels in that layout so far), -1 = lane- /
 * row-gather kernel, > 0 = LDS-staged ("a3_staged" value + 1 / 1);  max_unique = largest number of distinct source cells one
 * tile of the lists in use references (0 without lists). */
int mpg_handle_kernel_choice(mpg_handle rh, int *cell_fast_kernel, int *lev_fast_kernel, int *max_unique);
/* Locality of the source cells as the staged Regrid kernels see them, from the tile lists in use (error before the first
 * staged Regrid of the handle): the tile shape in target points; reuse = 3 * n_dst / (sum over the tiles of their
 * distinct source cells) -- how often a staged value is used; line_fill = the fraction of every 128-byte line of a
 * cell-fast float64 field touched by a tile that the tile actually uses (1 = its cells are dense runs of consecutive
 * ids, as on a row-numbered regional mesh; towards 1/16 = scattered ids: the mesh's cell numbering has no locality and
 * the level-fast (file-order) layout, which gathers whole rows, is the better route).  Diagnostics only; the reference
 * has no counterpart (ESMF hides its route handle). */
int mpg_handle_tile_stats(mpg_handle rh, int *tile_nx, int *tile_ny, double *reuse, double *line_fill);
/* Pole terms of a Grid -> Grid handle on a periodic (monopole) grid.  Destination points inside a pole cap
 * (triangle pole / A / B of the first or last CENTER row) carry, besides the A and B entries reported by
 * mpg_handle_get_weights, a weight on the pole node; the pole's value is the mean of the `row_len` sources
 * starting at `src_row_start`, i.e. ESMF's factor list holds row_len entries of w_pole/row_len for it.
 * mpg_handle_pole_count: n_points = 0 for every other handle.  mpg_handle_get_pole: arrays [n_points] (host);
 * entries with w_pole == 0 are destination points of the two candidate rows that are not in a cap. */
int mpg_handle_pole_count(mpg_handle rh, int64_t *n_points, int *row_len);
int mpg_handle_get_pole(mpg_handle rh, int32_t *dst_id_host, int32_t *src_row_start_host, double *w_pole_host);
/* dual (Delaunay) triangles of a mesh: tri_host [nVertices][3], 0-based cell ids or -1 */
int mpg_mesh_get_triangles(mpg_mesh mesh, int32_t *tri_host);

/* Multi-GPU (ESMF's per-Regrid source exchange): sorted unique source ids the handle
 * references.  Call with ids_host == NULL to get the count.  mpg_handle_localize rewrites the handle's
 * indices to positions in that list so that Regrid reads a compact [nlev][n_unique] halo buffer. */
int mpg_handle_unique_sources(mpg_handle rh, int64_t *n_unique, int32_t *ids_host);
/* Source window of a mesh: a host that holds, for every source field, only the contiguous id range [first, first + count)
 * its target rows reference -- one driver image per GPU reading just that range of every variable (MPAS variables are
 * [nCells][nLevels]: a cell range is one byte range of the file), where the reference has every rank read everything
 * (input_data.F90:645) and ESMF redistributes.  mpg_handle_source_range reports the global ids [first, end) a Mesh -> Grid
 * handle references (first == end: none).  mpg_mesh_set_source_window then declares the window for one mesh location:
 * every handle of that mesh and location -- existing (in use or parked in the cache) and future -- indexes its sources
 * relative to `first`, and Regrid reads source slabs of `count` ids ([nlev][count] / [count][nlev]); mpg_handle_info
 * reports n_src = count.  A handle that references a source outside the window fails the call.  Unlike the two calls
 * below the handles stay in the Store cache.  The whole mesh (first 0, count nCells / nVertices) resets it. */
int mpg_handle_source_range(mpg_handle rh, int64_t *first, int64_t *end);
int mpg_mesh_set_source_window(mpg_mesh mesh, int meshloc, int64_t first, int64_t count);
/* Both re-index the handle IN PLACE and detach it from the Store cache; a handle that is shared (mpg_regrid_store returned
 * the same pointer twice: refcount 2) is refused with MPG_ERR_INVALID_ARG -- release the other reference first. */
int mpg_handle_localize(mpg_handle rh);
/* Halo in "range" form (spatially banded cell numbering): subtract `base` from every source index and
 * declare the local source extent n_local, so that Regrid reads a [nlev][n_local] buffer holding the
 * global cell range [base, base + n_local).  Fails if any referenced id falls outside that range. */
int mpg_handle_rebase(mpg_handle rh, int64_t base, int64_t n_local);
/* dst[k][i] = src[k][ids[i]] (pack owned cells for the halo exchange); all device pointers */
int mpg_pack_dev(const double *src_dev, int64_t n_src, int nlev, const int32_t *ids_dev, int64_t n_ids,
                 double *dst_dev, void *hip_stream);

/* ---- several GPUs: one process per GPU, RCCL underneath (ESMF's source exchange and ESMF_FieldGather) -------------------
 * Target rows are sharded over the ranks (regDecomp = (/1, npets/), model_grid.F90:693) and source cells owned in contiguous
 * id blocks (model_grid.F90:423-438, 2428-2441); a rank whose rows reference cells of another rank's block receives them in
 * ONE grouped ncclSend / ncclRecv exchange per field batch -- what ESMF does inside every ESMF_FieldRegrid (interp.F90:134...).
 * No torch, no MPI: the ranks meet through `id_file` (rank 0 writes the RCCL unique id there; a path all ranks see, fresh per
 * run), the library loads librccl on first use.  A host that READS its sources from files needs none of this: it reads the
 * window its rows reference (mpg_mesh_set_source_window above); these verbs serve hosts whose source fields are produced or
 * held partitioned on the devices (bench.py --gpus N, a coupled model).
 *   mpg_comm_init / _destroy / _info      communicator of this process (its GPU = the one of mpg_init)
 *   mpg_comm_allgather                    small host-side metadata, rank order (blocking)
 *   mpg_halo_build                        the schedule for one Mesh -> Grid handle whose grid is this rank's row block:
 *                                         asks the handle for its source ids, agrees on the ownership blocks with the other
 *                                         ranks (ownership 0: block boundaries in the middle of the overlap of neighbouring
 *                                         ranks' needs -- only the strip along a row-block boundary travels; 1: equal blocks
 *                                         like the reference's para_range) and RE-INDEXES THE HANDLE IN PLACE to the local
 *                                         source space of n_local ids (range form: the covering id range, own block in place
 *                                         at own_pos; compact form for arbitrary numbering: the sorted needed ids) -- the
 *                                         handle leaves the Store cache, as with mpg_handle_rebase / _localize
 *   mpg_halo_exchange_dev                 one exchange of nrows rows (nfields * nlev): pack, grouped send / recv, unpack,
 *                                         enqueued on the stream; buffers allocated at the first call of a batch size
 *   mpg_gather_rows                       ESMF_FieldGather (write_data.F90:1006-1453): row blocks -> the whole field on root
 *   mpg_halo_plan_host                    the schedule as a pure function of every rank's needed ids (tests, diagnostics) */
/* The id file and its deadlines: rank 0 removes whatever is at `id_file`, writes {magic, launch tag = hash of the
 * environment's MPASSIT_RUN_ID (0 without), wall-clock time, nranks, id} and removes the file again once ncclCommInitRank
 * has returned; the other ranks ignore a file with another tag / nranks or written more than MPG_COMM_STALE_S (300) seconds
 * before they loaded the library.  Several ranks WITHOUT MPASSIT_RUN_ID are refused (MPG_ERR_INVALID_ARG: an untagged launch cannot
 * tell its file from one a crashed launch left behind); MPG_COMM_ALLOW_UNTAGGED=1 overrides, a reader then takes only a file at most
 * 30 s older than itself whose bytes are unchanged one second later.  Waiting for the file, ncclCommInitRank (run on a helper thread) and the blocking
 * all-gathers give up after MPG_COMM_TIMEOUT_S (120) seconds with MPG_ERR_TIMEOUT: a dead or missing peer is an error exit,
 * never a hang.  mpg_comm_idfile_verdict is that acceptance rule as a pure function (NULL = accepted; CPU tests). */
typedef struct mpg_comm_s *mpg_comm;
typedef struct mpg_halo_s *mpg_halo;
int mpg_comm_init(int rank, int nranks, const char *id_file, mpg_comm *out);
const char *mpg_comm_idfile_verdict(const void *bytes, int64_t nbytes, int nranks, uint64_t tag, int64_t reader_loaded_ns, double stale_s);
int mpg_comm_destroy(mpg_comm comm);
int mpg_comm_info(mpg_comm comm, int *rank, int *nranks);
int mpg_comm_allgather(mpg_comm comm, const void *send_host, int64_t nbytes, void *recv_host);
int mpg_halo_build(mpg_comm comm, mpg_handle rh, int64_t n_cells_global, int ownership, mpg_halo *out);
/* The same for a partition of the source cells that is the CALLER's (owned form): owned_ids_host = this rank's sorted unique global
 * ids, any shape -- the model's own decomposition of a coupled run (MPAS partitions its cells with a graph partitioner, not in id
 * blocks), or one that follows the target rows of a mesh without banded numbering: bench.py gives every cell to the lowest rank whose
 * rows reference it, so that only the overlap of neighbouring row blocks travels (equal id blocks of a Morton-numbered global mesh
 * would move 7/8 of all referenced values at 8 ranks).  The ranks' lists must be disjoint and cover every cell some rank references
 * (else MPG_ERR_INVALID_ARG).  The handle is re-indexed to the rank's sorted needed ids (n_local of them, as in the compact form);
 * mpg_halo_info reports mode 2 and own = {0, n_owned}; own_dev of mpg_halo_exchange_dev is [nrows][own_ld >= n_owned] in the order
 * of owned_ids_host.  What a peer sends arrives in id order and is scattered to its positions among the needed ids. */
int mpg_halo_build_owned(mpg_comm comm, mpg_handle rh, int64_t n_cells_global, const int32_t *owned_ids_host, int64_t n_owned, mpg_halo *out);
/* mode 0 range / 1 compact / 2 owned; own = the global id block [own[0], own[1]) this rank holds; base = global id of local index 0
 * (range form); own_pos = where the own block sits in the local space (range form); sent / received per row = elements that
 * cross the links per exchanged row.  Any pointer may be NULL. */
int mpg_halo_info(mpg_halo halo, int *mode, int64_t *n_local, int64_t *own, int64_t *base, int64_t *own_pos, int64_t *sent_per_row,
                  int64_t *received_per_row);
/* own_dev: nrows rows of the own block, row stride own_ld elements; local_dev: [nrows][n_local], filled in place.
 * elem_bytes: 4 or 8 for sources held cell-fast ([level][cell], input_data.F90:653-655: nrows = nfields * nlev), or the bytes of
 * one whole source row for sources held in MPAS file order ([cell][level], input_data.F90:630,645: nrows = nfields, elem_bytes =
 * nlev * 4 for float32, nlev * 8 for float64; any multiple of 4) -- in range form a neighbour's strip of a file-order slab is
 * ONE contiguous byte range per field, no pack step at all.
 * Range form: own_dev may be the in-place view (local_dev + own_pos[0] elements, own_ld = n_local) -- the own
 * data is then where Regrid reads it and only the neighbours' strips move -- or a separate buffer, which is copied to
 * its place first; any other overlap with local_dev is refused.  Compact form: own_dev is always a separate buffer. */
int mpg_halo_exchange_dev(mpg_halo halo, const void *own_dev, int64_t own_ld, void *local_dev, int nrows, int elem_bytes, void *hip_stream);
int mpg_halo_destroy(mpg_halo halo);
/* dst[k][i] = src[k * ld + ids[i]], nrows rows of elements of elem_bytes bytes (as above): the pack step of the compact halo
 * form on its own, for a host that runs the exchange through its own transport (mpassit_amd/dist.py on torch.distributed). */
int mpg_pack_rows_dev(const void *src_dev, int64_t ld, int nrows, int elem_bytes, const int32_t *ids_dev, int64_t n_ids, void *dst_dev, void *hip_stream);
/* REHEARSAL ON ONE GPU (tests and tools; never a production path).  RCCL refuses two ranks on one card, but it accepts
 * ncclSend / ncclRecv with peer == self inside one group.  mpg_comm_virtual makes virtual rank v_rank of v_nranks on top of
 * `real`, the ONE-rank communicator of the process; every virtual rank is driven by its own host thread and makes the calls a
 * real rank makes (mpg_comm_allgather, mpg_halo_build, mpg_halo_exchange_dev, mpg_gather_rows).  Each collective step is a
 * rendezvous of those threads; the last to arrive issues, for all of them, the very RCCL calls the real ranks would issue --
 * same entry points, same device pointers and byte counts from each rank's own schedule, ordered against each rank's stream
 * by events -- with every peer mapped to rank 0 of the real communicator.  Two ranks whose schedules disagree (a send that meets a receive
 * of another size) fail the step.  What it cannot show is bytes crossing xGMI.  A thread that never arrives ends the others'
 * wait with MPG_ERR_TIMEOUT after MPG_COMM_TIMEOUT_S.  mpg_comm_virtual_stats: groups, sends, receives and all-gathers the
 * group has really put through RCCL (any pointer may be NULL).  Destroy the virtual ranks before `real`. */
int mpg_comm_virtual(mpg_comm real, int v_rank, int v_nranks, mpg_comm *out);
int mpg_comm_virtual_stats(mpg_comm comm, int64_t *groups, int64_t *sends, int64_t *recvs, int64_t *allgathers);
/* rows_dev: this rank's [nlev][j1 - j0][nx] block of an [nlev][ny][nx] field; dst_dev (root only): the whole field */
int mpg_gather_rows(mpg_comm comm, const void *rows_dev, int64_t j0, int64_t j1, int64_t nx, int64_t ny, int nlev, int elem_bytes,
                    void *dst_dev, int root, void *hip_stream);
/* needed[q]: rank q's sorted unique needed ids (n_needed[q] of them).  Outputs for `rank` (arrays of nranks entries unless
 * noted): send_count; send_a = start of the range inside the own block (range form, -1 in compact form); recv_a / recv_b =
 * destination range in the local space; compact form: send_ids_flat (capacity send_ids_cap) + send_ids_off [nranks + 1]. */
int mpg_halo_plan_host(int rank, int nranks, int64_t n_cells, int ownership, const int64_t *n_needed, const int32_t *const *needed, int *mode,
                       int64_t *n_local, int64_t *own, int64_t *base, int64_t *own_pos, int64_t *send_count, int64_t *send_a, int64_t *recv_a,
                       int64_t *recv_b, int32_t *send_ids_flat, int64_t send_ids_cap, int64_t *send_ids_off);
/* the owned form's schedule as a pure function: per peer q (self included) send_flat[send_off[q] .. send_off[q + 1]) = offsets into
 * `rank`'s owned list in the order sent, recv_flat[recv_off[q] ..) = destination positions in its local space in the order received */
int mpg_halo_plan_owned_host(int rank, int nranks, const int64_t *n_needed, const int32_t *const *needed, const int64_t *n_owned,
                             const int32_t *const *owned, int64_t *n_local, int32_t *send_flat, int64_t send_cap, int64_t *send_off, int32_t *recv_flat,
                             int64_t recv_cap, int64_t *recv_off);

/* kernel-selection knobs for benchmarking and the A/B tests (defaults are the tuned production values; DESIGN.md s4.1 has
 * the measurements behind every default).  They select among kernels that produce identical bits:
 *   "a3_staged"   cell-fast 3-point Regrid: -1 per-handle choice (default), -2 lane-gather kernel, 0 / 1 / 2 the LDS-staged
 *                 kernel on 64x8-point tiles / 64x16 on 256 threads / 64x16 on 512 threads
 *   "lf_variant"  level-fast (file-order) 3-point Regrid: -1 per-handle choice (default), 0 row gather on linear aligned
 *                 tiles, 1 LDS-staged in 16-level chunks, 2 row gather on grid-row tiles (the capacity fallback)
 *   "nn_variant"  nearest-neighbour search: 1 wave-cooperative (default), 0 one thread per point
 *   "field_band"  order of the (field, tile) work items of a bundle Regrid: -1 each kernel's own choice (default: bands of
 *                 1024 tiles for the level-fast row gather, field-major for the staged kernels), 0 field-major, n > 0 all
 *                 fields of a band of n tiles before the next band
 *   "store_boxes" Stores on a grid that knows its projection: 1 (default) candidates through the grid's index space, 0 the
 *                 hierarchical search always
 * and one that does NOT (it selects between two readings of ESMF's undocumented-here behaviour):
 *   "bilinear_linetype"   Mesh -> Grid bilinear Store: 0 (default) the target point meets the plane of its source triangle
 *                 along the ray from the sphere's centre; 1 along the plane's normal (ESMF_LINETYPE_CART read literally).
 *                 In force at mpg_regrid_store time; the two differ by O(h^2) of the triangle size
 *   "lfu_min_reuse_x10"   threshold (x10) of the per-handle level-fast choice (default 35)
 *   "lfu_npf"     row slots per thread of the staged level-fast kernel: 0 (default) = by the handle's longest tile list (2 / 4 / 8 / 16),
 *                 16 = the fixed shape of rounds 1-4 (A/B measurements; the results are the same bits)
 *   "lf_rows_store", "staged_store"   how results are STORED (never what: the same bits).  A result is [nlev][ny][nx]; level k's plane starts on a
 *                 128-byte line only when ny * nx is a multiple of 32 (float32) / 16 (float64) points, and the kernels' runs of 64 points of a plane
 *                 that does not have a partial line at either end.  Default 0: whole lines non-temporal, partial lines write-back (per lane;
 *                 the row gather's float32 results per level).  "lf_rows_store" 1 / 2 / 3: the file-order row gather plain / non-temporal / per lane;
 *                 "staged_store" 2: the staged cell-fast kernel non-temporal on every lane (what rounds 2-6a shipped).  A/B only
 *                 (profiles/r06_plane_alignment.md: 14-29 % on grids with an odd number of points per level)
 * Unknown keys and out-of-range values return MPG_ERR_INVALID_ARG. */
int mpg_tune(const char *key, int value);

/* timing of the last Store phases in ms (search build, search, finalize); any pointer may be NULL */
int mpg_handle_store_ms(mpg_handle rh, float *ms_total);
/* How the Store found its candidates: 0 = the hierarchical search (pyramid walk over the grid / BVH over the mesh);
 * 1 = through the grid's index space (a grid that knows its projection, mpg_grid_create_proj / mpg_grid_attach_proj, fine
 * enough for the claim to hold -- DESIGN.md s4.2); 2 = nearest only: index space for the points it can vouch for, the BVH for
 * the others.  The weights are the same bits either way; diagnostics only. */
int mpg_handle_store_path(mpg_handle rh, int *candidates);
/* Which data-dependent branches the Store of a Mesh -> Grid handle took: n values (up to 8; the rest 0) into stats_host.  [0] = the
 * store path above.  Then by method --
 *   bilinear:     [1] triangles handed to the wave-per-triangle rasteriser (no usable index near a pole / the projection's cut, or
 *                 more than 16 leaves of the pyramid), [2] triangles in all
 *   nearest:      [1] bin side in grid points (2 where the mesh is as fine as the grid .. 16), [2] bins, [3] points the bins could not
 *                 vouch for (settled by the tree), [4] cells on and around the grid the bin side was sized from
 *   conservative: [1] (cell, destination cell) pairs clipped, [2] polygons with more candidates than their 24-entry list (spilled
 *                 past it into their overflow slot, or counted and listed by a workgroup), [3] polygons whose index box a wavefront
 *                 enumerated (more than 128 box cells), [4] polygons the cooperative count pass walked the pyramid for, [5] lists
 *                 the list pass copied from the spill area instead of walking again, [6] vertex slots per polygon of the clip
 * Diagnostics (the tests use them to assert that a case built to force a branch did take it); the weights do not depend on them. */
int mpg_handle_store_stats(mpg_handle rh, int64_t *stats_host, int n);

/* Test hook: the library's own device-wide primitives (csrc/k_prims.hip; they took rocPRIM's place in round 5) on host arrays --
 * out_host[i] = in_host[0] + ... + in_host[i - 1] (int32, exclusive; out_host may be NULL) and *sum_host = the 64-bit sum (may be NULL).
 * The Stores use them on device counts; this entry exists so that a test can ask them directly at awkward sizes. */
int mpg_debug_scan_i32(const int32_t *in_host, int64_t n, int32_t *out_host, long long *sum_host);

#ifdef __cplusplus
}
#endif
#endif
