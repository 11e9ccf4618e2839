!> ISO_C_BINDING interface to libmpassit_amd.so (include/mpassit_amd.h) -- the thin Fortran<->HIP boundary
!! named by BASELINE.json's north_star.  Each interface replaces one ESMF verb of the reference:
!!   mpg_init / mpg_finalize       ESMF_Initialize / ESMF_Finalize          mpassit.F90:84,140
!!   mpg_mesh_create               ESMF_MeshCreate                          model_grid.F90:488-497
!!   mpg_grid_create               ESMF_GridCreate* + ESMF_GridAddCoord     model_grid.F90:684-728,736-1038
!!   mpg_regrid_store[_grid]       ESMF_Field[Bundle]RegridStore            interp.F90:123,207,...,437
!!   mpg_regrid                    ESMF_Field[Bundle]Regrid                 interp.F90:134,219,...,443
!!   mpg_handle_release            ESMF_FieldBundleRegridRelease            interp.F90:450-463
!!   mpg_rotate_winds              rotate_winds_cgrid                       interp.F90:689-749
!! Array conventions line up with the reference without any transposition:
!!   Fortran verticesOnCell(maxEdges,nCells) == C [nCells][maxEdges];  field(nCells,nz) == C [nz][nCells]
!!   (cell-fastest, input_data.F90:653-655);  target lat(i,j) == C [ny][nx];  dst(i,j,k) == C [nz][ny][nx].
module mpg
  use, intrinsic :: iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: MPG_SUCCESS = 0, MPG_ERR_UNSUPPORTED = 4
  integer(c_int), parameter :: MPG_REGRIDMETHOD_BILINEAR = 0, MPG_REGRIDMETHOD_CONSERVE = 1, MPG_REGRIDMETHOD_NEAREST_STOD = 2
  integer(c_int), parameter :: MPG_MESHLOC_ELEMENT = 0, MPG_MESHLOC_NODE = 1
  integer(c_int), parameter :: MPG_STAGGERLOC_CENTER = 0, MPG_STAGGERLOC_EDGE1 = 1, MPG_STAGGERLOC_EDGE2 = 2, &
                               MPG_STAGGERLOC_CORNER = 3
  integer(c_int), parameter :: MPG_LAYOUT_CELL_FAST = 0, MPG_LAYOUT_LEV_FAST = 1
  !> element type codes of mpg_regrid_typed[_dev]: float64 / float32, + MPG_TYPE_BE when the values are big-endian in
  !! memory (the bytes of a NetCDF classic variable, moved file <-> HBM untouched)
  integer(c_int), parameter :: MPG_TYPE_F64 = 0, MPG_TYPE_F32 = 1, MPG_TYPE_BE = 2
  integer(c_int), parameter :: MPG_GRID_PERIODIC_I = 1, MPG_GRID_NO_SOUTH_POLE = 2, MPG_GRID_NO_NORTH_POLE = 4
  integer(c_int), parameter :: MPG_PROJ_LATLON = 0, MPG_PROJ_LC = 1, MPG_PROJ_PS = 2, MPG_PROJ_MERC = 3
  !> struct mpg_proj (include/mpassit_amd.h): the arguments of push_source_projection (model_grid.F90:676-678)
  type, bind(C) :: mpg_proj
    integer(c_int) :: code
    real(c_double) :: known_lat, known_lon, known_x, known_y
    real(c_double) :: dx_m
    real(c_double) :: stand_lon, truelat1, truelat2
    real(c_double) :: dlat_deg, dlon_deg
  end type mpg_proj

  interface
    function mpg_init(device) bind(C, name="mpg_init") result(rc)
      import :: c_int
      integer(c_int), value :: device
      integer(c_int) :: rc
    end function mpg_init

    !> a run-time choice of the library by name (include/mpassit_amd.h: bilinear_linetype, node_fan_origin, grid_inside_tol_exp, ...)
    function mpg_tune_c(key, value) bind(C, name="mpg_tune") result(rc)
      import :: c_int, c_char
      character(kind=c_char), intent(in) :: key(*)
      integer(c_int), value :: value
      integer(c_int) :: rc
    end function mpg_tune_c

    function mpg_device_count(n) bind(C, name="mpg_device_count") result(rc)
      import :: c_int
      integer(c_int), intent(out) :: n
      integer(c_int) :: rc
    end function mpg_device_count

    function mpg_finalize() bind(C, name="mpg_finalize") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function mpg_finalize

    function mpg_last_error_c() bind(C, name="mpg_last_error") result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function mpg_last_error_c

    function mpg_mesh_create(nCells, nVertices, maxEdges, latCell, lonCell, latVertex, lonVertex, verticesOnCell, mesh) &
        bind(C, name="mpg_mesh_create") result(rc)
      import :: c_int, c_int64_t, c_int32_t, c_double, c_ptr
      integer(c_int64_t), value :: nCells, nVertices
      integer(c_int), value :: maxEdges
      real(c_double), intent(in) :: latCell(*), lonCell(*), latVertex(*), lonVertex(*)
      integer(c_int32_t), intent(in) :: verticesOnCell(*)
      type(c_ptr), intent(out) :: mesh
      integer(c_int) :: rc
    end function mpg_mesh_create

    !> ESMF_MeshCreate for ONE image of a job whose target rows are split over several GPUs: only the part of the mesh that
    !! `grid` (this image's row block) can see is brought to the device; ids stay global, weights are those of the whole mesh
    function mpg_mesh_create_window(nCells, nVertices, maxEdges, latCell, lonCell, latVertex, lonVertex, verticesOnCell, grid, mesh) &
        bind(C, name="mpg_mesh_create_window") result(rc)
      import :: c_int, c_int64_t, c_int32_t, c_double, c_ptr
      integer(c_int64_t), value :: nCells, nVertices
      integer(c_int), value :: maxEdges
      real(c_double), intent(in) :: latCell(*), lonCell(*), latVertex(*), lonVertex(*)
      integer(c_int32_t), intent(in) :: verticesOnCell(*)
      type(c_ptr), value :: grid
      type(c_ptr), intent(out) :: mesh
      integer(c_int) :: rc
    end function mpg_mesh_create_window

    function mpg_mesh_window_info(mesh, cell_first, cell_count, vertex_first, vertex_count, margin) bind(C, name="mpg_mesh_window_info") result(rc)
      import :: c_int, c_int64_t, c_double, c_ptr
      type(c_ptr), value :: mesh
      integer(c_int64_t), intent(out) :: cell_first, cell_count, vertex_first, vertex_count
      real(c_double), intent(out) :: margin
      integer(c_int) :: rc
    end function mpg_mesh_window_info

    function mpg_mesh_destroy(mesh) bind(C, name="mpg_mesh_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: mesh
      integer(c_int) :: rc
    end function mpg_mesh_destroy

    function mpg_grid_create(nx, ny, periodic_i, lon_center, lat_center, lon_corner, lat_corner, lon_edge1, lat_edge1, &
                             lon_edge2, lat_edge2, grid) bind(C, name="mpg_grid_create") result(rc)
      import :: c_int, c_double, c_ptr
      integer(c_int), value :: nx, ny, periodic_i
      real(c_double), intent(in) :: lon_center(*), lat_center(*), lon_corner(*), lat_corner(*)
      real(c_double), intent(in) :: lon_edge1(*), lat_edge1(*), lon_edge2(*), lat_edge2(*)
      type(c_ptr), intent(out) :: grid
      integer(c_int) :: rc
    end function mpg_grid_create

    function mpg_grid_destroy(grid) bind(C, name="mpg_grid_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: grid
      integer(c_int) :: rc
    end function mpg_grid_destroy

    function mpg_regrid_store(src, src_meshloc, dst, dst_staggerloc, regridmethod, rh) &
        bind(C, name="mpg_regrid_store") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: src, dst
      integer(c_int), value :: src_meshloc, dst_staggerloc, regridmethod
      type(c_ptr), intent(out) :: rh
      integer(c_int) :: rc
    end function mpg_regrid_store

    function mpg_regrid_store_grid(grid, src_staggerloc, dst_staggerloc, regridmethod, rh) &
        bind(C, name="mpg_regrid_store_grid") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: grid
      integer(c_int), value :: src_staggerloc, dst_staggerloc, regridmethod
      type(c_ptr), intent(out) :: rh
      integer(c_int) :: rc
    end function mpg_regrid_store_grid

    !> the two Stores STARTED on the library's worker thread (include/mpassit_amd.h): the matching mpg_regrid_store[_grid] collects them
    function mpg_regrid_store_begin(src, src_meshloc, dst, dst_staggerloc, regridmethod) bind(C, name="mpg_regrid_store_begin") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: src, dst
      integer(c_int), value :: src_meshloc, dst_staggerloc, regridmethod
      integer(c_int) :: rc
    end function mpg_regrid_store_begin
    function mpg_regrid_store_grid_begin(grid, src_staggerloc, dst_staggerloc, regridmethod) bind(C, name="mpg_regrid_store_grid_begin") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: grid
      integer(c_int), value :: src_staggerloc, dst_staggerloc, regridmethod
      integer(c_int) :: rc
    end function mpg_regrid_store_grid_begin

    function mpg_regrid(rh, src, src_layout, nlev, nfields, dst) bind(C, name="mpg_regrid") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: rh
      real(c_double), intent(in) :: src(*)
      integer(c_int), value :: src_layout, nlev, nfields
      real(c_double), intent(inout) :: dst(*)
      integer(c_int) :: rc
    end function mpg_regrid

    !> Regrid on host buffers of the files' own types: src_type / dst_type = MPG_TYPE_F32 for real(c_float) arrays (MPAS
    !! history variables are NF90_FLOAT), MPG_TYPE_F64 for real(c_double); float64 arithmetic, dst = regrid(src)*scale + offset.
    function mpg_regrid_typed(rh, src, src_f32, src_layout, nlev, nfields, dst, dst_f32, scale, offset) &
      bind(C, name="mpg_regrid_typed") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: rh, src, dst
      integer(c_int), value :: src_f32, src_layout, nlev, nfields, dst_f32
      real(c_double), value :: scale, offset
      integer(c_int) :: rc
    end function mpg_regrid_typed

    !> the same on device pointers (mpg_dev_alloc), enqueued on `stream` (c_null_ptr = default stream); src_f32 / dst_f32
    !! are MPG_TYPE_* codes: + MPG_TYPE_BE for the raw bytes of a NetCDF classic variable
    function mpg_regrid_typed_dev(rh, src, src_f32, src_layout, nlev, nfields, dst, dst_f32, scale, offset, stream) &
      bind(C, name="mpg_regrid_typed_dev") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: rh, src, dst, stream
      integer(c_int), value :: src_f32, src_layout, nlev, nfields, dst_f32
      real(c_double), value :: scale, offset
      integer(c_int) :: rc
    end function mpg_regrid_typed_dev

    !> ESMF_FieldBundleRegrid over the SEPARATE device arrays of a bundle's fields (interp.F90:240-254): one launch for all of
    !! them, offsets(nfields) = the epilogue offset of each field
    function mpg_regrid_bundle_typed_dev(rh, nfields, src, src_f32, src_layout, nlev, dst, dst_f32, scale, offsets, stream) &
      bind(C, name="mpg_regrid_bundle_typed_dev") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: rh, stream
      integer(c_int), value :: nfields, src_f32, src_layout, nlev, dst_f32
      type(c_ptr), intent(in) :: src(*), dst(*)
      real(c_double), value :: scale
      real(c_double), intent(in) :: offsets(*)
      integer(c_int) :: rc
    end function mpg_regrid_bundle_typed_dev

    !> the same on separate HOST arrays: every field of the bundle through one upload / Regrid / download pipeline
    function mpg_regrid_bundle_typed(rh, nfields, src, src_f32, src_layout, nlev, dst, dst_f32, scale, offsets) &
      bind(C, name="mpg_regrid_bundle_typed") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: rh
      integer(c_int), value :: nfields, src_f32, src_layout, nlev, dst_f32
      type(c_ptr), intent(in) :: src(*), dst(*)
      real(c_double), value :: scale
      real(c_double), intent(in) :: offsets(*)
      integer(c_int) :: rc
    end function mpg_regrid_bundle_typed

    function mpg_rotate_winds_dev(npts, nlev, cosa, sina, u, v, stream) bind(C, name="mpg_rotate_winds_dev") result(rc)
      import :: c_int, c_int64_t, c_ptr
      integer(c_int64_t), value :: npts
      integer(c_int), value :: nlev
      type(c_ptr), value :: cosa, sina, u, v, stream
      integer(c_int) :: rc
    end function mpg_rotate_winds_dev

    !> interp.F90:291-328 in one pass: rotate_winds_cgrid + UMASS -> U(EDGE1) + VMASS -> V(EDGE2) (include/mpassit_amd.h)
    function mpg_wind_destagger_dev(rh_edge1, rh_edge2, cosa, sina, umass, vmass, nlev, u, v, dst_type, umass_rot, vmass_rot, stream) &
      bind(C, name="mpg_wind_destagger_dev") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: rh_edge1, rh_edge2, cosa, sina, umass, vmass, u, v, umass_rot, vmass_rot, stream
      integer(c_int), value :: nlev, dst_type
      integer(c_int) :: rc
    end function mpg_wind_destagger_dev
    !> the same chain on HOST arrays: the mass winds cross the link once, only U and V (and, on request, the rotated mass winds -- which
    !! may be the input arrays: rotate_winds_cgrid's in-place result) come back (include/mpassit_amd.h)
    function mpg_wind_destagger(rh_edge1, rh_edge2, cosa, sina, umass, vmass, nlev, u, v, dst_type, umass_rot, vmass_rot) &
      bind(C, name="mpg_wind_destagger") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: rh_edge1, rh_edge2, cosa, sina, umass, vmass, u, v, umass_rot, vmass_rot
      integer(c_int), value :: nlev, dst_type
      integer(c_int) :: rc
    end function mpg_wind_destagger

    !> device buffers for fields that stay in HBM between the input and the output file
    function mpg_dev_alloc(nbytes, dev) bind(C, name="mpg_dev_alloc") result(rc)
      import :: c_int, c_int64_t, c_ptr
      integer(c_int64_t), value :: nbytes
      type(c_ptr), intent(out) :: dev
      integer(c_int) :: rc
    end function mpg_dev_alloc
    function mpg_dev_free(dev) bind(C, name="mpg_dev_free") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: dev
      integer(c_int) :: rc
    end function mpg_dev_free
    function mpg_dev_upload(dst_dev, src_host, nbytes) bind(C, name="mpg_dev_upload") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: dst_dev
      type(*), dimension(*), intent(in) :: src_host
      integer(c_int64_t), value :: nbytes
      integer(c_int) :: rc
    end function mpg_dev_upload

    !> bytes [offset, offset + nbytes) of a file <-> device memory, untouched (NetCDF classic variables: ncio_var_extent)
    function mpg_file_to_dev_c(path, offset, nbytes, dst_dev, stream) bind(C, name="mpg_file_to_dev") result(rc)
      import :: c_int, c_int64_t, c_ptr, c_char
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int64_t), value :: offset, nbytes
      type(c_ptr), value :: dst_dev, stream
      integer(c_int) :: rc
    end function mpg_file_to_dev_c
    function mpg_dev_to_file_c(path, offset, nbytes, src_dev, stream) bind(C, name="mpg_dev_to_file") result(rc)
      import :: c_int, c_int64_t, c_ptr, c_char
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int64_t), value :: offset, nbytes
      type(c_ptr), value :: src_dev, stream
      integer(c_int) :: rc
    end function mpg_dev_to_file_c
    !> global source ids [first, last) a Mesh -> Grid handle references; source window of a mesh location
    function mpg_handle_source_range(rh, first, last) bind(C, name="mpg_handle_source_range") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: rh
      integer(c_int64_t), intent(out) :: first, last
      integer(c_int) :: rc
    end function mpg_handle_source_range
    function mpg_mesh_set_source_window(mesh, meshloc, first, count) bind(C, name="mpg_mesh_set_source_window") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: mesh
      integer(c_int), value :: meshloc
      integer(c_int64_t), value :: first, count
      integer(c_int) :: rc
    end function mpg_mesh_set_source_window
    !> several GPUs, one image per GPU, RCCL underneath (include/mpassit_amd.h): communicator, halo schedule of a route
    !! handle + its exchange, ESMF_FieldGather.  id_file: a path all images see, fresh per run (image 0 writes the id).
    function mpg_comm_init(rank, nranks, id_file, comm) bind(C, name="mpg_comm_init") result(rc)
      import :: c_int, c_char, c_ptr
      integer(c_int), value :: rank, nranks
      character(kind=c_char), intent(in) :: id_file(*)
      type(c_ptr), intent(out) :: comm
      integer(c_int) :: rc
    end function mpg_comm_init
    function mpg_comm_destroy(comm) bind(C, name="mpg_comm_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: comm
      integer(c_int) :: rc
    end function mpg_comm_destroy
    function mpg_halo_build(comm, rh, n_cells_global, ownership, halo) bind(C, name="mpg_halo_build") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: comm, rh
      integer(c_int64_t), value :: n_cells_global
      integer(c_int), value :: ownership
      type(c_ptr), intent(out) :: halo
      integer(c_int) :: rc
    end function mpg_halo_build
    !> the same for the CALLER's own partition of the source cells (a coupled model's decomposition): owned_ids = this image's sorted
    !! unique global cell ids, 0-based; own_dev of mpg_halo_exchange_dev is then [nrows][own_ld >= n_owned] in that order
    function mpg_halo_build_owned(comm, rh, n_cells_global, owned_ids, n_owned, halo) bind(C, name="mpg_halo_build_owned") result(rc)
      import :: c_int, c_int32_t, c_int64_t, c_ptr
      type(c_ptr), value :: comm, rh
      integer(c_int64_t), value :: n_cells_global, n_owned
      integer(c_int32_t), intent(in) :: owned_ids(*)
      type(c_ptr), intent(out) :: halo
      integer(c_int) :: rc
    end function mpg_halo_build_owned
    function mpg_halo_info(halo, mode, n_local, own, base, own_pos, sent_per_row, received_per_row) bind(C, name="mpg_halo_info") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: halo
      integer(c_int), intent(out) :: mode
      integer(c_int64_t), intent(out) :: n_local, own(2), base, own_pos(2), sent_per_row, received_per_row
      integer(c_int) :: rc
    end function mpg_halo_info
    function mpg_halo_exchange_dev(halo, own_dev, own_ld, local_dev, nrows, elem_bytes, stream) bind(C, name="mpg_halo_exchange_dev") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: halo, own_dev, local_dev, stream
      integer(c_int64_t), value :: own_ld
      integer(c_int), value :: nrows, elem_bytes
      integer(c_int) :: rc
    end function mpg_halo_exchange_dev
    function mpg_halo_destroy(halo) bind(C, name="mpg_halo_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: halo
      integer(c_int) :: rc
    end function mpg_halo_destroy
    function mpg_gather_rows(comm, rows_dev, j0, j1, nx, ny, nlev, elem_bytes, dst_dev, root, stream) bind(C, name="mpg_gather_rows") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: comm, rows_dev, dst_dev, stream
      integer(c_int64_t), value :: j0, j1, nx, ny
      integer(c_int), value :: nlev, elem_bytes, root
      integer(c_int) :: rc
    end function mpg_gather_rows
    !> blocks until mpg_init's helper thread is done (call before a global-mode stream capture that follows mpg_init at once)
    function mpg_warmup_wait() bind(C, name="mpg_warmup_wait") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function mpg_warmup_wait
    !> which data-dependent branches a Store took (layout by method: include/mpassit_amd.h); diagnostics
    function mpg_handle_store_stats(rh, stats, n) bind(C, name="mpg_handle_store_stats") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: rh
      integer(c_int64_t), intent(out) :: stats(*)
      integer(c_int), value :: n
      integer(c_int) :: rc
    end function mpg_handle_store_stats
    function mpg_bswap_dev(buf, n, elem_size, stream) bind(C, name="mpg_bswap_dev") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: buf, stream
      integer(c_int64_t), value :: n
      integer(c_int), value :: elem_size
      integer(c_int) :: rc
    end function mpg_bswap_dev

    !> writer epilogues on device-resident float64 fields (write_data.F90:1339-1475), float32 results, stored big-endian
    !! (as the output file holds them) when dst_be /= 0
    function mpg_post_cast_dev(src, n, scale, offset, dst, dst_be, stream) bind(C, name="mpg_post_cast_dev") result(rc)
      import :: c_int, c_int64_t, c_double, c_ptr
      type(c_ptr), value :: src, dst, stream
      integer(c_int64_t), value :: n
      real(c_double), value :: scale, offset
      integer(c_int), value :: dst_be
      integer(c_int) :: rc
    end function mpg_post_cast_dev
    function mpg_post_layer_mean_dev(src, nlevp1, npts, dst, dst_be, stream) bind(C, name="mpg_post_layer_mean_dev") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: src, dst, stream
      integer(c_int), value :: nlevp1, dst_be
      integer(c_int64_t), value :: npts
      integer(c_int) :: rc
    end function mpg_post_layer_mean_dev
    function mpg_post_ptop_dev(p_hyd, nlev, npts, ptop, stream) bind(C, name="mpg_post_ptop_dev") result(rc)
      import :: c_int, c_int64_t, c_double, c_ptr
      type(c_ptr), value :: p_hyd, stream
      integer(c_int), value :: nlev
      integer(c_int64_t), value :: npts
      real(c_double), intent(out) :: ptop
      integer(c_int) :: rc
    end function mpg_post_ptop_dev
    function mpg_post_ptop_parts_dev(p_hyd, nlev, npts, vmax, candmin, has_cand, stream) bind(C, name="mpg_post_ptop_parts_dev") result(rc)
      import :: c_int, c_int64_t, c_double, c_ptr
      type(c_ptr), value :: p_hyd, stream
      integer(c_int), value :: nlev
      integer(c_int64_t), value :: npts
      real(c_double), intent(out) :: vmax, candmin
      integer(c_int), intent(out) :: has_cand
      integer(c_int) :: rc
    end function mpg_post_ptop_parts_dev

    function mpg_handle_release(rh) bind(C, name="mpg_handle_release") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: rh
      integer(c_int) :: rc
    end function mpg_handle_release

    function mpg_rotate_winds(npts, nlev, cosa, sina, u, v) bind(C, name="mpg_rotate_winds") result(rc)
      import :: c_int, c_int64_t, c_double
      integer(c_int64_t), value :: npts
      integer(c_int), value :: nlev
      real(c_double), intent(in) :: cosa(*), sina(*)
      real(c_double), intent(inout) :: u(*), v(*)
      integer(c_int) :: rc
    end function mpg_rotate_winds

    function mpg_handle_info(rh, n_src, n_dst, nx_dst, ny_dst, nnz_per_row, nnz) bind(C, name="mpg_handle_info") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: rh
      integer(c_int64_t), intent(out) :: n_src, n_dst, nnz
      integer(c_int), intent(out) :: nx_dst, ny_dst, nnz_per_row
      integer(c_int) :: rc
    end function mpg_handle_info

    !> Target grid straight from the projection (define_target_grid_params' host loops on the GPU, model_grid.F90:736-1038).
    !> a grid made from coordinate arrays that are rows row0 .. of the grid `proj` describes (an image's row block): its Stores
    !! search through the inverse projection; refused when the projection does not reproduce the grid's own points
    function mpg_grid_attach_proj(grid, proj, row0) bind(C, name="mpg_grid_attach_proj") result(rc)
      import :: c_int, c_ptr, mpg_proj
      type(c_ptr), value :: grid
      type(mpg_proj), intent(in) :: proj
      integer(c_int), value :: row0
      integer(c_int) :: rc
    end function mpg_grid_attach_proj

    function mpg_grid_create_proj(proj, nx, ny, periodic_i, grid) bind(C, name="mpg_grid_create_proj") result(rc)
      import :: c_int, c_ptr, mpg_proj
      type(mpg_proj), intent(in) :: proj
      integer(c_int), value :: nx, ny, periodic_i
      type(c_ptr), intent(out) :: grid
      integer(c_int) :: rc
    end function mpg_grid_create_proj

    function mpg_grid_get_coords(grid, staggerloc, lon, lat) bind(C, name="mpg_grid_get_coords") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: grid
      integer(c_int), value :: staggerloc
      real(c_double), intent(out) :: lon(*), lat(*)
      integer(c_int) :: rc
    end function mpg_grid_get_coords

    function mpg_grid_get_rotang(grid, cosa, sina) bind(C, name="mpg_grid_get_rotang") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: grid
      real(c_double), intent(out) :: cosa(*), sina(*)
      integer(c_int) :: rc
    end function mpg_grid_get_rotang
    !> cos / sin(alpha) where mpg_grid_create_proj computed them: device pointers owned by the grid ([ny][nx], CENTER)
    function mpg_grid_rotang_dev(grid, cosa_dev, sina_dev) bind(C, name="mpg_grid_rotang_dev") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: grid
      type(c_ptr), intent(out) :: cosa_dev, sina_dev
      integer(c_int) :: rc
    end function mpg_grid_rotang_dev

    function mpg_grid_get_mapfac(grid, staggerloc, mapfac) bind(C, name="mpg_grid_get_mapfac") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: grid
      integer(c_int), value :: staggerloc
      real(c_double), intent(out) :: mapfac(*)
      integer(c_int) :: rc
    end function mpg_grid_get_mapfac
  end interface

contains

  !> Message of the last failing call (mpg_last_error).
  function mpg_last_error() result(msg)
    character(len=:), allocatable :: msg
    character(kind=c_char), pointer :: s(:)
    type(c_ptr) :: p
    integer :: n
    p = mpg_last_error_c()
    msg = ""
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [1024])
    n = 0
    do while (n < 1024)
      if (s(n + 1) == c_null_char) exit
      n = n + 1
    end do
    allocate (character(len=n) :: msg)
    if (n > 0) msg = transfer(s(1:n), msg)
  end function mpg_last_error

  !> Same contract as the reference's error_handler (utils.F90:16-33): print and abort with code 999 on any
  !! non-zero rc -- every ESMF rc in interp.F90 is checked this way (e.g. :130-131).
  integer(c_int) function mpg_file_to_dev(path, offset, nbytes, dst_dev) result(rc)
    character(len=*), intent(in) :: path
    integer(c_int64_t), intent(in) :: offset, nbytes
    type(c_ptr), intent(in) :: dst_dev
    rc = mpg_file_to_dev_c(trim(path)//c_null_char, offset, nbytes, dst_dev, c_null_ptr)
  end function mpg_file_to_dev

  integer(c_int) function mpg_dev_to_file(path, offset, nbytes, src_dev) result(rc)
    character(len=*), intent(in) :: path
    integer(c_int64_t), intent(in) :: offset, nbytes
    type(c_ptr), intent(in) :: src_dev
    rc = mpg_dev_to_file_c(trim(path)//c_null_char, offset, nbytes, src_dev, c_null_ptr)
  end function mpg_dev_to_file

  !> MPASSIT_TUNE="key=value[,key=value...]": the library's run-time choices for a site whose ESMF comparison (tools/esmf_pin.py compare)
  !! named another setting than the default -- e.g. MPASSIT_TUNE="bilinear_linetype=1,node_fan_origin=-1".  Called once after mpg_init.
  subroutine mpg_apply_tune_env()
    character(len=512) :: txt
    character(len=64) :: key
    integer :: i, j, k, val, ios
    call get_environment_variable("MPASSIT_TUNE", txt)
    i = 1
    do while (i <= len_trim(txt))
      j = index(txt(i:), ",")
      if (j == 0) then
        j = len_trim(txt) + 1
      else
        j = i + j - 1
      end if
      k = index(txt(i:j - 1), "=")
      if (k < 2) then
        print *, "- FATAL ERROR: MPASSIT_TUNE wants key=value[,key=value...]: ", trim(txt)
        error stop 998
      end if
      key = adjustl(txt(i:i + k - 2))
      read (txt(i + k:j - 1), *, iostat=ios) val
      if (ios /= 0) then
        print *, "- FATAL ERROR: MPASSIT_TUNE: not an integer in ", txt(i:j - 1)
        error stop 998
      end if
      call mpg_check(mpg_tune_c(trim(key)//c_null_char, int(val, c_int)), "MPASSIT_TUNE "//txt(i:j - 1))
      print '(a,a,a,i0)', " - LIBRARY CHOICE ", trim(key), " = ", val
      i = j + 1
    end do
  end subroutine mpg_apply_tune_env

  subroutine mpg_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    if (rc == MPG_SUCCESS) return
    print *, "- FATAL ERROR: "
    write (*, '(A)') trim(where)//": "//mpg_last_error()
    print *, "- IOSTAT IS: ", rc
    error stop 999
  end subroutine mpg_check

end module mpg
