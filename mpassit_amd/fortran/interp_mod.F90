!> Fortran mirror of the reference's `interp` module (interp.F90:92-749) on top of the HIP C-ABI.
!! Same public entry point (interp_data), same order of operations and method fall-through; every ESMF
!! Store/Regrid/Release becomes the mpg_* call named in mpg_mod.F90.  Field containers replace the ESMF
!! field bundles of model_grid.F90:154-237.
module model_data
  use, intrinsic :: iso_c_binding
  use program_setup, only: dp
  implicit none
  public
  type field_t
    character(len=50) :: name = "", tname = ""
    integer :: nlev = 1
    integer :: stagger = 0                      ! MPG_STAGGERLOC_* of the destination
    real(dp), allocatable :: src(:)             ! MPAS file order: (nlev, nCells) level-fastest, or (nCells) for 2-D
    real(c_float), allocatable :: src4(:)       ! the same when the file variable is NF90_FLOAT (kept single: half the
                                                ! host memory and PCIe bytes; the Regrid kernel widens in its loads)
    real(dp), allocatable :: dst(:)             ! (nx, ny, nlev) i-fastest
    real(c_float), allocatable :: dst4(:)       ! the same as the NF90_FLOAT the output file holds (f32_out), instead of dst
    ! device-resident flow (dev_flow): the field lives in HBM from the input file to the output file
    type(c_ptr) :: src_dev = c_null_ptr, dst_dev = c_null_ptr
    logical :: src_is_f32 = .false., dst_is_f32 = .false.
    !> the device buffer holds the file's big-endian bytes as they are (NetCDF classic): the Regrid reads / writes them so
    logical :: src_is_be = .false., dst_is_be = .false.
    integer(c_int64_t) :: n_dst_elems = 0
  end type field_t
  type bundle_t
    integer :: n = 0
    type(field_t), allocatable :: f(:)
  end type bundle_t

  type(c_ptr) :: input_grid = c_null_ptr, target_grid_h = c_null_ptr   ! ESMF_Mesh / ESMF_Grid stand-ins
  integer :: nCells_input = 0, nVert_input = 0, maxEdges_input = 0, nz_input = 0, nzp1_input = 0, nsoil_input = 0
  type(bundle_t), target :: diag_bundle, hist_2d_patch, hist_2d_cons, hist_2d_nstd, hist_3d_nz, hist_3d_nzp1, hist_3d_vert, hist_soil
  type(field_t), target :: hgt, u_field, v_field, umass, vmass
  real(dp), allocatable :: zs_input(:)          ! soil layer depths of the grid file (ZS of the output, write_data.F90:1130)
  integer :: do_u_interp = 0, do_v_interp = 0, u10_ind = 0, v10_ind = 0
  !> the output is a NetCDF file (every variable NF90_FLOAT, write_data.F90:587-980): fields nothing else needs in
  !! float64 come back from the Regrid as float32 with the writer's affine post-op fused (T - 300, write_data.F90:1343)
  logical :: f32_out = .false.
  !> NetCDF classic files in and out: variables travel file <-> GPU as raw bytes (mpg_file_to_dev / mpg_dev_to_file,
  !! byte order turned on the device) and every field stays in device buffers (mpg_dev_alloc) in between; the host
  !! arrays of field_t are not used at all
  logical :: dev_flow = .false.
  !> device flow: the source window of this image (mpg_mesh_set_source_window) -- the cell / vertex ids [win0, win0 + winn)
  !! its target rows reference; only those rows of every variable are read from the input files.  winn < 0: whole mesh
  integer(c_int64_t) :: win0_cell = 0, winn_cell = -1, win0_vert = 0, winn_vert = -1

contains

  !> fields the host still works on in float64 after the Regrid: rotated winds (interp.F90:138-140,291-293),
  !! PHB (Z_C and PHB*9.81, write_data.F90:1406-1418), P_HYD (P_TOP, :1362-1371)
  logical function keeps_r8(f)
    type(field_t), intent(in) :: f
    keeps_r8 = .not. f32_out .or. trim(f%name) == 'u10' .or. trim(f%name) == 'v10' .or. &
               trim(f%name) == 'uReconstructZonal' .or. trim(f%name) == 'uReconstructMeridional' .or. &
               trim(f%tname) == 'PHB' .or. trim(f%tname) == 'P_HYD'
  end function keeps_r8

  !> a -> b without copying the arrays; a is left empty
  subroutine move_field(a, b)
    type(field_t), intent(inout) :: a, b
    b%name = a%name; b%tname = a%tname; b%nlev = a%nlev; b%stagger = a%stagger
    if (allocated(b%src)) deallocate (b%src)
    if (allocated(b%src4)) deallocate (b%src4)
    if (allocated(b%dst)) deallocate (b%dst)
    if (allocated(b%dst4)) deallocate (b%dst4)
    if (allocated(a%src)) call move_alloc(a%src, b%src)
    if (allocated(a%src4)) call move_alloc(a%src4, b%src4)
    if (allocated(a%dst)) call move_alloc(a%dst, b%dst)
    if (allocated(a%dst4)) call move_alloc(a%dst4, b%dst4)
    b%src_dev = a%src_dev; b%dst_dev = a%dst_dev; a%src_dev = c_null_ptr; a%dst_dev = c_null_ptr
    b%src_is_f32 = a%src_is_f32; b%dst_is_f32 = a%dst_is_f32; b%n_dst_elems = a%n_dst_elems
    b%src_is_be = a%src_is_be; b%dst_is_be = a%dst_is_be
  end subroutine move_field
end module model_data

module interp
  use, intrinsic :: iso_c_binding
  use mpg
  use model_data
  use program_setup, only: dp, interp_diag, interp_hist, proj_code, PROJ_LC, i_target, j_target, wrf_mod_vars, je_lo, je_hi, ny_ext
  use target_grid, only: cosa, sina
  implicit none
  private
  public :: interp_data
  type(c_ptr) :: cosa_dev = c_null_ptr, sina_dev = c_null_ptr
  logical :: rotang_owned = .false.     ! cosa_dev / sina_dev were allocated here (not the grid's own arrays)

contains

  subroutine interp_data()
    call begin_stores()
    if (interp_diag) call interp_diag_data()
    if (interp_hist) call interp_hist_data()
  end subroutine interp_data

  !> Every RegridStore of the run (interp.F90:123, 207-437) started now, in the order of use, on the library's worker thread: the
  !! reference stores each weight set in front of the Regrids that use it; they are independent, so the conservative, nearest and
  !! destaggering weights build while the bilinear Regrids are running.  The Store calls further down collect them (same weights).
  subroutine begin_stores()
    if (.not. (interp_diag .or. interp_hist)) return
    call mpg_check(mpg_regrid_store_begin(input_grid, MPG_MESHLOC_ELEMENT, target_grid_h, MPG_STAGGERLOC_CENTER, MPG_REGRIDMETHOD_BILINEAR), &
                   "IN FieldBundleRegridStore (begin)")
    if (.not. interp_hist) return
    if (do_u_interp == 1) call mpg_check(mpg_regrid_store_grid_begin(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE1, &
                                                                      MPG_REGRIDMETHOD_BILINEAR), "IN FieldRegridStore (begin)")
    if (do_v_interp == 1) call mpg_check(mpg_regrid_store_grid_begin(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE2, &
                                                                      MPG_REGRIDMETHOD_BILINEAR), "IN FieldRegridStore (begin)")
    if (hist_3d_vert%n > 0) call mpg_check(mpg_regrid_store_begin(input_grid, MPG_MESHLOC_NODE, target_grid_h, MPG_STAGGERLOC_CENTER, &
                                                                   MPG_REGRIDMETHOD_BILINEAR), "IN FieldBundleRegridStore (begin)")
    if (hist_2d_cons%n > 0) call mpg_check(mpg_regrid_store_begin(input_grid, MPG_MESHLOC_ELEMENT, target_grid_h, MPG_STAGGERLOC_CENTER, &
                                                                   MPG_REGRIDMETHOD_CONSERVE), "IN FieldBundleRegridStore (begin)")
    if (hist_2d_nstd%n > 0) call mpg_check(mpg_regrid_store_begin(input_grid, MPG_MESHLOC_ELEMENT, target_grid_h, MPG_STAGGERLOC_CENTER, &
                                                                   MPG_REGRIDMETHOD_NEAREST_STOD), "IN FieldBundleRegridStore (begin)")
  end subroutine begin_stores

  !> ESMF_FieldBundleRegrid: every field of the bundle through one route handle.
  !! 3-D sources arrive in MPAS file order (level fastest): the GPU kernel fuses the transpose the reference
  !! does on the host (input_data.F90:653-655).
  subroutine regrid_bundle(rh, b)
    type(c_ptr), intent(in) :: rh
    type(bundle_t), intent(inout) :: b
    integer :: i
    if (dev_flow .and. b%n > 1) then
      call regrid_bundle_dev(rh, b)
      return
    end if
    if (b%n > 1) then
      call regrid_bundle_host(rh, b)
      return
    end if
    do i = 1, b%n
      call regrid_field(rh, b%f(i))
    end do
  end subroutine regrid_bundle

  !> Host arrays: the fields of the bundle that agree in level count and element types go through ONE pipeline
  !! (mpg_regrid_bundle_typed): the upload of one field, the Regrid of the one before and the download of the one before that
  !! overlap, where a file-order field handed over alone runs the three steps one after the other.
  subroutine regrid_bundle_host(rh, b)
    type(c_ptr), intent(in) :: rh
    type(bundle_t), intent(inout), target :: b
    integer(c_int64_t) :: n_src, n_dst, nnz
    integer(c_int) :: nxd, nyd, npr, layout
    integer :: i, j, ng
    logical :: done(b%n), r8(b%n), s4(b%n)
    type(c_ptr) :: sp(b%n), dp(b%n)
    real(c_double) :: offs(b%n)
    type(field_t), pointer :: f, g
    call mpg_check(mpg_handle_info(rh, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
    do i = 1, b%n
      f => b%f(i)
      if (allocated(f%dst)) deallocate (f%dst)
      if (allocated(f%dst4)) deallocate (f%dst4)
      r8(i) = keeps_r8(f)
      s4(i) = allocated(f%src4)
      if (r8(i)) then
        allocate (f%dst(n_dst*f%nlev))
      else
        allocate (f%dst4(n_dst*f%nlev))
      end if
    end do
    done = .false.
    do i = 1, b%n
      if (done(i)) cycle
      f => b%f(i)
      ng = 0
      do j = i, b%n
        g => b%f(j)
        if (done(j) .or. g%nlev /= f%nlev .or. (s4(j) .neqv. s4(i)) .or. (r8(j) .neqv. r8(i))) cycle
        ng = ng + 1
        if (s4(j)) then
          sp(ng) = c_loc(g%src4)
        else
          sp(ng) = c_loc(g%src)
        end if
        offs(ng) = 0.0_c_double
        if (r8(j)) then
          dp(ng) = c_loc(g%dst)
        else
          dp(ng) = c_loc(g%dst4)
          if (wrf_mod_vars .and. trim(g%tname) == 'T') offs(ng) = -300.0_c_double
        end if
        done(j) = .true.
      end do
      layout = MPG_LAYOUT_LEV_FAST
      if (f%nlev == 1) layout = MPG_LAYOUT_CELL_FAST
      call mpg_check(mpg_regrid_bundle_typed(rh, int(ng, c_int), sp, merge(1_c_int, 0_c_int, s4(i)), layout, int(f%nlev, c_int), dp, &
                                             merge(0_c_int, 1_c_int, r8(i)), 1.0_c_double, offs), "IN FieldBundleRegrid "//trim(f%name))
    end do
  end subroutine regrid_bundle_host

  !> Device flow: the fields of the bundle that agree in level count and element types go through ONE Regrid
  !! (mpg_regrid_bundle_typed_dev over their separate device arrays, per-field epilogue offsets), as the reference's
  !! ESMF_FieldBundleRegrid does; the same bits as field-by-field calls, one launch instead of b%n.
  subroutine regrid_bundle_dev(rh, b)
    type(c_ptr), intent(in) :: rh
    type(bundle_t), intent(inout), target :: b
    integer(c_int64_t) :: n_src, n_dst, nnz
    integer(c_int) :: nxd, nyd, npr, layout
    integer :: i, j, ng
    logical :: done(b%n)
    type(c_ptr) :: sp(b%n), dp(b%n)
    real(c_double) :: offs(b%n)
    type(field_t), pointer :: f, g
    call mpg_check(mpg_handle_info(rh, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
    do i = 1, b%n                                   ! what regrid_field does before its call
      f => b%f(i)
      if (allocated(f%dst)) deallocate (f%dst)
      if (allocated(f%dst4)) deallocate (f%dst4)
      if (c_associated(f%dst_dev)) call mpg_check(mpg_dev_free(f%dst_dev), "IN dev_free")
      f%dst_is_f32 = .not. keeps_r8(f)
      f%dst_is_be = f%dst_is_f32
      f%n_dst_elems = n_dst*f%nlev
      call mpg_check(mpg_dev_alloc(f%n_dst_elems*merge(4, 8, f%dst_is_f32), f%dst_dev), "IN dev_alloc "//trim(f%name))
    end do
    done = .false.
    do i = 1, b%n
      if (done(i)) cycle
      f => b%f(i)
      ng = 0
      do j = i, b%n
        g => b%f(j)
        if (done(j) .or. g%nlev /= f%nlev .or. (g%src_is_f32 .neqv. f%src_is_f32) .or. (g%src_is_be .neqv. f%src_is_be) .or. &
            (g%dst_is_f32 .neqv. f%dst_is_f32) .or. (g%dst_is_be .neqv. f%dst_is_be)) cycle
        ng = ng + 1
        sp(ng) = g%src_dev
        dp(ng) = g%dst_dev
        offs(ng) = 0.0_c_double
        if (g%dst_is_f32 .and. wrf_mod_vars .and. trim(g%tname) == 'T') offs(ng) = -300.0_c_double
        done(j) = .true.
      end do
      layout = MPG_LAYOUT_LEV_FAST
      if (f%nlev == 1) layout = MPG_LAYOUT_CELL_FAST
      call mpg_check(mpg_regrid_bundle_typed_dev(rh, int(ng, c_int), sp, elem_type(f%src_is_f32, f%src_is_be), layout, int(f%nlev, c_int), &
                                                 dp, elem_type(f%dst_is_f32, f%dst_is_be), 1.0_c_double, offs, c_null_ptr), &
                     "IN FieldBundleRegrid "//trim(f%name))
    end do
    do i = 1, b%n                                   ! the sources are not needed again
      call mpg_check(mpg_dev_free(b%f(i)%src_dev), "IN dev_free")
      b%f(i)%src_dev = c_null_ptr
    end do
  end subroutine regrid_bundle_dev

  subroutine regrid_field(rh, f)
    type(c_ptr), intent(in) :: rh
    type(field_t), intent(inout), target :: f
    integer(c_int64_t) :: n_src, n_dst, nnz
    integer(c_int) :: nxd, nyd, npr, layout, src_f32
    type(c_ptr) :: sp
    real(c_double) :: offset
    call mpg_check(mpg_handle_info(rh, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
    if (allocated(f%dst)) deallocate (f%dst)
    if (allocated(f%dst4)) deallocate (f%dst4)
    layout = MPG_LAYOUT_LEV_FAST
    if (f%nlev == 1) layout = MPG_LAYOUT_CELL_FAST
    if (dev_flow) then
      if (c_associated(f%dst_dev)) call mpg_check(mpg_dev_free(f%dst_dev), "IN dev_free")
      f%dst_is_f32 = .not. keeps_r8(f)
      f%dst_is_be = f%dst_is_f32                 ! NF90_FLOAT results go straight to the file: produced as the file stores them
      f%n_dst_elems = n_dst*f%nlev
      call mpg_check(mpg_dev_alloc(f%n_dst_elems*merge(4, 8, f%dst_is_f32), f%dst_dev), "IN dev_alloc "//trim(f%name))
      offset = 0.0_c_double
      if (f%dst_is_f32 .and. wrf_mod_vars .and. trim(f%tname) == 'T') offset = -300.0_c_double
      call mpg_check(mpg_regrid_typed_dev(rh, f%src_dev, elem_type(f%src_is_f32, f%src_is_be), layout, int(f%nlev, c_int), 1_c_int, &
                                          f%dst_dev, elem_type(f%dst_is_f32, f%dst_is_be), 1.0_c_double, offset, c_null_ptr), &
                     "IN FieldRegrid "//trim(f%name))
      call mpg_check(mpg_dev_free(f%src_dev), "IN dev_free")      ! the source is not needed again
      f%src_dev = c_null_ptr
      return
    end if
    if (allocated(f%src4)) then
      sp = c_loc(f%src4); src_f32 = 1
    else
      sp = c_loc(f%src); src_f32 = 0
    end if
    if (keeps_r8(f)) then
      allocate (f%dst(n_dst*f%nlev))
      call mpg_check(mpg_regrid_typed(rh, sp, src_f32, layout, int(f%nlev, c_int), 1_c_int, c_loc(f%dst), 0_c_int, &
                                      1.0_c_double, 0.0_c_double), "IN FieldRegrid "//trim(f%name))
    else
      offset = 0.0_c_double
      if (wrf_mod_vars .and. trim(f%tname) == 'T') offset = -300.0_c_double
      allocate (f%dst4(n_dst*f%nlev))
      call mpg_check(mpg_regrid_typed(rh, sp, src_f32, layout, int(f%nlev, c_int), 1_c_int, c_loc(f%dst4), 1_c_int, &
                                      1.0_c_double, offset), "IN FieldRegrid "//trim(f%name))
    end if
  end subroutine regrid_field

  integer(c_int) function elem_type(is_f32, is_be)
    logical, intent(in) :: is_f32, is_be
    elem_type = merge(MPG_TYPE_F32, MPG_TYPE_F64, is_f32) + merge(MPG_TYPE_BE, 0_c_int, is_be)
  end function elem_type

  subroutine interp_diag_data()
    type(c_ptr) :: rh_patch
    print *, "- CREATE DIAG BUNDLE REGRID ROUTEHANDLE"
    call mpg_check(mpg_regrid_store(input_grid, MPG_MESHLOC_ELEMENT, target_grid_h, MPG_STAGGERLOC_CENTER, &
                                    MPG_REGRIDMETHOD_BILINEAR, rh_patch), "IN FieldBundleRegridStore")
    print *, "- REGRID DIAG FIELDS "
    call regrid_bundle(rh_patch, diag_bundle)
    call mpg_check(mpg_handle_release(rh_patch), "IN FieldRegridRelease")
    if (u10_ind > 0 .and. v10_ind > 0 .and. proj_code == PROJ_LC) then
      call rotate_winds_cgrid(diag_bundle%f(u10_ind), diag_bundle%f(v10_ind))
    end if
  end subroutine interp_diag_data

  subroutine interp_hist_data()
    type(c_ptr) :: rh_patch, rh_cons, rh_nstd, rh_stag, rh_soil
    integer(c_int) :: method
    logical :: have_cons, have_nstd
    ! the reference leaves `method` undefined when no 2-D bilinear field is listed (interp.F90:203-204,
    ! SURVEY App. C2); it is always BILINEAR here
    method = MPG_REGRIDMETHOD_BILINEAR
    print *, "- CREATE HIST BUNDLE PATCH REGRID ROUTEHANDLE"
    call mpg_check(mpg_regrid_store(input_grid, MPG_MESHLOC_ELEMENT, target_grid_h, MPG_STAGGERLOC_CENTER, method, rh_patch), &
                   "IN FieldBundleRegridStore")
    if (hist_2d_patch%n > 0) call regrid_bundle(rh_patch, hist_2d_patch)
    print *, "- PATCH REGRID HGT FIELD "
    call regrid_field(rh_patch, hgt)
    if (hist_3d_nz%n > 0) call regrid_bundle(rh_patch, hist_3d_nz)
    if (do_u_interp == 1) call regrid_to(rh_patch, u_field, umass)
    if (do_v_interp == 1) call regrid_to(rh_patch, v_field, vmass)
    if (dev_flow .and. (do_u_interp == 1 .or. do_v_interp == 1)) then
      ! device-resident fields: rotation and both destaggerings in ONE pass over the mass winds (mpg_wind_destagger_dev); a pair of
      ! handles it does not take (MPG_ERR_UNSUPPORTED) leaves the three calls below to do the work
      if (wind_chain_fused()) go to 100
    else if (do_u_interp == 1 .or. do_v_interp == 1) then
      ! host arrays: the same chain through ONE upload of the mass winds (mpg_wind_destagger) instead of three calls that move them
      ! up twice and down once more
      if (wind_chain_fused_host()) go to 100
    end if
    if (do_u_interp == 1 .and. do_v_interp == 1 .and. proj_code == PROJ_LC) call rotate_winds_cgrid(umass, vmass)
    if (do_u_interp == 1) then          ! UMASS(CENTER) -> U(EDGE1), interp.F90:295-311
      call mpg_check(mpg_regrid_store_grid(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE1, method, rh_stag), &
                     "IN FieldRegridStore")
      call destagger(rh_stag, umass, u_field)
      call mpg_check(mpg_handle_release(rh_stag), "IN FieldRegridRelease")
    end if
    if (do_v_interp == 1) then          ! VMASS(CENTER) -> V(EDGE2), interp.F90:313-328
      call mpg_check(mpg_regrid_store_grid(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE2, method, rh_stag), &
                     "IN FieldRegridStore")
      call destagger(rh_stag, vmass, v_field)
      call mpg_check(mpg_handle_release(rh_stag), "IN FieldRegridRelease")
    end if
100 continue
    if (dev_flow) then                  ! UMASS / VMASS are not output variables of the NetCDF file
      if (c_associated(umass%dst_dev)) call mpg_check(mpg_dev_free(umass%dst_dev), "IN dev_free")
      if (c_associated(vmass%dst_dev)) call mpg_check(mpg_dev_free(vmass%dst_dev), "IN dev_free")
      umass%dst_dev = c_null_ptr; vmass%dst_dev = c_null_ptr
    end if
    if (hist_3d_nzp1%n > 0) call regrid_bundle(rh_patch, hist_3d_nzp1)
    if (hist_3d_vert%n > 0) then        ! node-located sources (vorticity), interp.F90:350-366
      print *, "- CREATE HIST BUNDLE VERT BILINEAR REGRID ROUTEHANDLE"
      call mpg_check(mpg_regrid_store(input_grid, MPG_MESHLOC_NODE, target_grid_h, MPG_STAGGERLOC_CENTER, method, rh_stag), &
                     "IN FieldBundleRegridStore")
      call regrid_bundle(rh_stag, hist_3d_vert)
      call mpg_check(mpg_handle_release(rh_stag), "IN FieldRegridRelease")
    end if
    rh_soil = rh_patch
    have_cons = hist_2d_cons%n > 0
    have_nstd = hist_2d_nstd%n > 0
    if (have_cons) then
      print *, "- CREATE HIST BUNDLE CONSERVATIVE REGRID ROUTEHANDLE"
      method = MPG_REGRIDMETHOD_CONSERVE
      call mpg_check(mpg_regrid_store(input_grid, MPG_MESHLOC_ELEMENT, target_grid_h, MPG_STAGGERLOC_CENTER, method, rh_cons), &
                     "IN FieldBundleRegridStore")
      call regrid_bundle(rh_cons, hist_2d_cons)
      rh_soil = rh_cons
    end if
    if (have_nstd) then
      print *, "- CREATE HIST BUNDLE NSTD REGRID ROUTEHANDLE"
      method = MPG_REGRIDMETHOD_NEAREST_STOD
      call mpg_check(mpg_regrid_store(input_grid, MPG_MESHLOC_ELEMENT, target_grid_h, MPG_STAGGERLOC_CENTER, method, rh_nstd), &
                     "IN FieldBundleRegridStore")
      call regrid_bundle(rh_nstd, hist_2d_nstd)
      rh_soil = rh_nstd
    end if
    ! soil: "whatever method is current" (interp.F90:436-447, SURVEY App. C3)
    if (hist_soil%n > 0) call regrid_bundle(rh_soil, hist_soil)
    print *, "- CALL FieldRegridRelease."
    call mpg_check(mpg_handle_release(rh_patch), "IN FieldRegridRelease")
    if (have_cons) call mpg_check(mpg_handle_release(rh_cons), "IN FieldRegridRelease")
    if (have_nstd) call mpg_check(mpg_handle_release(rh_nstd), "IN FieldRegridRelease")
  end subroutine interp_hist_data

  !> cos / sin(alpha) of this image's row block on the device: the grid's own arrays when mpg_grid_create_proj built it there (one
  !! image, grid from the namelist's projection: no upload at all), else uploaded once from the host's copies
  subroutine rotang_on_device()
    integer(c_int64_t) :: npts
    type(c_ptr) :: ca, sa
    if (c_associated(cosa_dev)) return
    if (ny_ext == j_target) then
      if (mpg_grid_rotang_dev(target_grid_h, ca, sa) == MPG_SUCCESS) then
        cosa_dev = ca; sina_dev = sa; rotang_owned = .false.
        return
      end if
    end if
    npts = int(i_target, c_int64_t)*int(ny_ext, c_int64_t)  ! this image's row block (all rows with one image)
    call mpg_check(mpg_dev_alloc(npts*8, cosa_dev), "IN dev_alloc")
    call mpg_check(mpg_dev_alloc(npts*8, sina_dev), "IN dev_alloc")
    call mpg_check(mpg_dev_upload(cosa_dev, cosa(:, je_lo:je_hi), npts*8), "IN dev_upload")
    call mpg_check(mpg_dev_upload(sina_dev, sina(:, je_lo:je_hi), npts*8), "IN dev_upload")
    rotang_owned = .true.
  end subroutine rotang_on_device

  !> interp.F90:291-328 on device-resident mass winds: rotate_winds_cgrid (PROJ_LC, both components) and the two Grid -> Grid
  !! Store / Regrid pairs as one kernel pass; U / V come out as the file stores them (NF90_FLOAT, big-endian), bit-identical to
  !! the three calls.  .false.: the library does not take this pair of handles -- nothing has been done.
  logical function wind_chain_fused()
    type(c_ptr) :: rh_u, rh_v, ca, sa
    integer(c_int64_t) :: n_src, n_dst, nnz, npts
    integer(c_int) :: nxd, nyd, npr, rc, nlev
    logical :: rot
    wind_chain_fused = .false.
    rh_u = c_null_ptr; rh_v = c_null_ptr; ca = c_null_ptr; sa = c_null_ptr
    rot = do_u_interp == 1 .and. do_v_interp == 1 .and. proj_code == PROJ_LC
    nlev = int(merge(umass%nlev, vmass%nlev, do_u_interp == 1), c_int)
    if (do_u_interp == 1) then          ! the mass winds must be the float64 fields regrid_to left (keeps_r8)
      if (umass%dst_is_f32 .or. umass%dst_is_be) return
    end if
    if (do_v_interp == 1) then
      if (vmass%dst_is_f32 .or. vmass%dst_is_be) return
    end if
    if (do_u_interp == 1) then
      call mpg_check(mpg_regrid_store_grid(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE1, MPG_REGRIDMETHOD_BILINEAR, rh_u), &
                     "IN FieldRegridStore")
    end if
    if (do_v_interp == 1) then
      call mpg_check(mpg_regrid_store_grid(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE2, MPG_REGRIDMETHOD_BILINEAR, rh_v), &
                     "IN FieldRegridStore")
    end if
    if (rot) then
      call rotang_on_device()
      ca = cosa_dev; sa = sina_dev
    end if
    if (do_u_interp == 1) then
      call mpg_check(mpg_handle_info(rh_u, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
      if (c_associated(u_field%dst_dev)) call mpg_check(mpg_dev_free(u_field%dst_dev), "IN dev_free")
      u_field%dst_is_f32 = .true.; u_field%dst_is_be = .true.; u_field%n_dst_elems = n_dst*nlev
      call mpg_check(mpg_dev_alloc(u_field%n_dst_elems*4, u_field%dst_dev), "IN dev_alloc")
    end if
    if (do_v_interp == 1) then
      call mpg_check(mpg_handle_info(rh_v, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
      if (c_associated(v_field%dst_dev)) call mpg_check(mpg_dev_free(v_field%dst_dev), "IN dev_free")
      v_field%dst_is_f32 = .true.; v_field%dst_is_be = .true.; v_field%n_dst_elems = n_dst*nlev
      call mpg_check(mpg_dev_alloc(v_field%n_dst_elems*4, v_field%dst_dev), "IN dev_alloc")
    end if
    rc = mpg_wind_destagger_dev(rh_u, rh_v, ca, sa, umass%dst_dev, vmass%dst_dev, nlev, u_field%dst_dev, v_field%dst_dev, &
                                MPG_TYPE_F32 + MPG_TYPE_BE, c_null_ptr, c_null_ptr, c_null_ptr)
    if (c_associated(rh_u)) call mpg_check(mpg_handle_release(rh_u), "IN FieldRegridRelease")
    if (c_associated(rh_v)) call mpg_check(mpg_handle_release(rh_v), "IN FieldRegridRelease")
    if (rc == MPG_ERR_UNSUPPORTED) return
    call mpg_check(rc, "IN wind_destagger")
    wind_chain_fused = .true.
  end function wind_chain_fused

  !> interp.F90:291-328 on HOST arrays (the NetCDF-4 / raw-file flows): mpg_wind_destagger uploads the float64 mass winds once in chunks of
  !! levels and brings back U and V as the output holds them (NF90_FLOAT for a NetCDF file, float64 else); the rotated mass winds come
  !! back into umass / vmass only where they are output variables (the raw format: not f32_out).  Bit-identical to the three calls.
  !! .false.: the library does not take this pair of handles, or the mass winds are not float64 host arrays -- nothing has been done.
  logical function wind_chain_fused_host()
    type(c_ptr) :: rh_u, rh_v, ca, sa, pum, pvm, pu, pv, pur, pvr
    integer(c_int64_t) :: n_src, n_dst, nnz
    integer(c_int) :: nxd, nyd, npr, rc, nlev
    logical :: rot
    wind_chain_fused_host = .false.
    rh_u = c_null_ptr; rh_v = c_null_ptr; ca = c_null_ptr; sa = c_null_ptr
    pum = c_null_ptr; pvm = c_null_ptr; pu = c_null_ptr; pv = c_null_ptr; pur = c_null_ptr; pvr = c_null_ptr
    rot = do_u_interp == 1 .and. do_v_interp == 1 .and. proj_code == PROJ_LC
    nlev = int(merge(umass%nlev, vmass%nlev, do_u_interp == 1), c_int)
    if (do_u_interp == 1) then
      if (.not. allocated(umass%dst)) return
    end if
    if (do_v_interp == 1) then
      if (.not. allocated(vmass%dst)) return
    end if
    if (do_u_interp == 1) then
      call mpg_check(mpg_regrid_store_grid(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE1, MPG_REGRIDMETHOD_BILINEAR, rh_u), &
                     "IN FieldRegridStore")
      call mpg_check(mpg_handle_info(rh_u, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
      if (allocated(u_field%dst)) deallocate (u_field%dst)
      if (allocated(u_field%dst4)) deallocate (u_field%dst4)
      if (f32_out) then
        allocate (u_field%dst4(n_dst*nlev)); pu = c_loc(u_field%dst4)
      else
        allocate (u_field%dst(n_dst*nlev)); pu = c_loc(u_field%dst)
      end if
      pum = c_loc(umass%dst)
    end if
    if (do_v_interp == 1) then
      call mpg_check(mpg_regrid_store_grid(target_grid_h, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE2, MPG_REGRIDMETHOD_BILINEAR, rh_v), &
                     "IN FieldRegridStore")
      call mpg_check(mpg_handle_info(rh_v, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
      if (allocated(v_field%dst)) deallocate (v_field%dst)
      if (allocated(v_field%dst4)) deallocate (v_field%dst4)
      if (f32_out) then
        allocate (v_field%dst4(n_dst*nlev)); pv = c_loc(v_field%dst4)
      else
        allocate (v_field%dst(n_dst*nlev)); pv = c_loc(v_field%dst)
      end if
      pvm = c_loc(vmass%dst)
    end if
    if (rot) then
      ca = c_loc(cosa); sa = c_loc(sina)
      if (.not. f32_out) then        ! UMASS / VMASS are output variables of the raw format: rotated in place, as rotate_winds_cgrid leaves them
        pur = pum; pvr = pvm
      end if
    end if
    rc = mpg_wind_destagger(rh_u, rh_v, ca, sa, pum, pvm, nlev, pu, pv, merge(MPG_TYPE_F32, MPG_TYPE_F64, f32_out), pur, pvr)
    if (c_associated(rh_u)) call mpg_check(mpg_handle_release(rh_u), "IN FieldRegridRelease")
    if (c_associated(rh_v)) call mpg_check(mpg_handle_release(rh_v), "IN FieldRegridRelease")
    if (rc == MPG_ERR_UNSUPPORTED) return
    call mpg_check(rc, "IN wind_destagger")
    wind_chain_fused_host = .true.
  end function wind_chain_fused_host

  !> mesh field `src` -> CENTER-stagger field `dst` (uReconstructZonal -> UMASS, interp.F90:256-289)
  subroutine regrid_to(rh, src, dst)
    type(c_ptr), intent(in) :: rh
    type(field_t), intent(inout) :: src
    type(field_t), intent(inout) :: dst
    call regrid_field(rh, src)
    dst%nlev = src%nlev
    if (dev_flow) then
      dst%dst_dev = src%dst_dev; src%dst_dev = c_null_ptr
      dst%dst_is_f32 = src%dst_is_f32; dst%dst_is_be = src%dst_is_be; dst%n_dst_elems = src%n_dst_elems
      return
    end if
    call move_alloc(src%dst, dst%dst)
  end subroutine regrid_to

  !> CENTER-stagger field -> EDGE stagger field through a Grid->Grid handle
  subroutine destagger(rh, mass, stag)
    type(c_ptr), intent(in) :: rh
    type(field_t), intent(in), target :: mass
    type(field_t), intent(inout), target :: stag
    integer(c_int64_t) :: n_src, n_dst, nnz
    integer(c_int) :: nxd, nyd, npr
    call mpg_check(mpg_handle_info(rh, n_src, n_dst, nxd, nyd, npr, nnz), "IN HandleInfo")
    if (allocated(stag%dst)) deallocate (stag%dst)
    if (allocated(stag%dst4)) deallocate (stag%dst4)
    if (dev_flow) then
      if (c_associated(stag%dst_dev)) call mpg_check(mpg_dev_free(stag%dst_dev), "IN dev_free")
      stag%dst_is_f32 = .true.; stag%dst_is_be = .true.
      stag%n_dst_elems = n_dst*mass%nlev
      call mpg_check(mpg_dev_alloc(stag%n_dst_elems*4, stag%dst_dev), "IN dev_alloc")
      call mpg_check(mpg_regrid_typed_dev(rh, mass%dst_dev, elem_type(mass%dst_is_f32, mass%dst_is_be), MPG_LAYOUT_CELL_FAST, &
                                          int(mass%nlev, c_int), 1_c_int, stag%dst_dev, MPG_TYPE_F32 + MPG_TYPE_BE, 1.0_c_double, 0.0_c_double, &
                                          c_null_ptr), "IN FieldRegrid")
      return
    end if
    if (f32_out) then
      allocate (stag%dst4(n_dst*mass%nlev))
      call mpg_check(mpg_regrid_typed(rh, c_loc(mass%dst), 0_c_int, MPG_LAYOUT_CELL_FAST, int(mass%nlev, c_int), 1_c_int, &
                                      c_loc(stag%dst4), 1_c_int, 1.0_c_double, 0.0_c_double), "IN FieldRegrid")
    else
      allocate (stag%dst(n_dst*mass%nlev))
      call mpg_check(mpg_regrid(rh, mass%dst, MPG_LAYOUT_CELL_FAST, int(mass%nlev, c_int), 1_c_int, stag%dst), "IN FieldRegrid")
    end if
  end subroutine destagger

  !> rotate_winds_cgrid (interp.F90:689-749) on the device
  subroutine rotate_winds_cgrid(u, v)
    type(field_t), intent(inout) :: u, v
    integer(c_int64_t) :: npts
    if (dev_flow) then
      npts = int(i_target, c_int64_t)*int(ny_ext, c_int64_t)  ! this image's row block (all rows with one image)
      call rotang_on_device()
      call mpg_check(mpg_rotate_winds_dev(npts, int(u%nlev, c_int), cosa_dev, sina_dev, u%dst_dev, v%dst_dev, c_null_ptr), &
                     "IN rotate_winds_cgrid")
      return
    end if
    call mpg_check(mpg_rotate_winds(int(i_target, c_int64_t)*int(j_target, c_int64_t), int(u%nlev, c_int), cosa, sina, u%dst, v%dst), &
                   "IN rotate_winds_cgrid")
  end subroutine rotate_winds_cgrid

end module interp
