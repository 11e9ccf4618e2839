!> Thin Fortran driver with the reference's user surface (mpassit.F90:22-146): `mpassit <namelist>`
!! (default fort.41), &config namelist, diaglist / histlist_2d / histlist_3d / histlist_soil in the CWD.
!! Phases in the reference's order (mpassit.F90:105-137): namelist -> target grid -> input grid ->
!! input data -> interp_data (HIP, through the C-ABI) -> write.  No MPI/ESMF: one process drives one GPU.
!! File I/O: NetCDF files -- the classic formats CDF-1/2/5 by this repository's own reader / writer, NetCDF-4 through libhdf5 where
!! libmpassit_ncio was built with it -- are recognised by their magic; anything else is taken as the MPGRAW1 raw container.  An
!! output_file ending in ".nc" is written as CDF-5 (MPASSIT_OUTPUT_FORMAT=netcdf4: as NetCDF-4, what the reference creates) with the
!! reference's dimension / variable names and post-ops (ncfiles_mod.F90).
!! Two data flows: with NetCDF at both ends every listed variable goes file -> GPU as raw bytes, stays in device buffers
!! through Regrid / rotation / destaggering / post-ops and goes GPU -> file the same way (dev_flow; no host array, no
!! host conversion); otherwise (raw container, or MPASSIT_HOST_ARRAYS set) fields live in host arrays like the
!! reference's and cross PCIe inside mpg_regrid / mpg_regrid_typed.  Both write the same bytes.
program mpassit
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int32, int64
  use mpg
  use program_setup
  use varlists
  use target_grid
  use rawio
  use model_data
  use interp
  use ncio
  use ncfiles
  implicit none
  character(len=500) :: nml_file
  character(len=19) :: valid_time = "0000-00-00_00:00:00"
  logical :: nc_in = .false.
  type(c_ptr) :: nf_in = c_null_ptr
  integer :: nargs, gpu
  integer(c_int) :: ngpu
  character(len=16) :: envbuf
  logical :: no_reserve, no_window
  integer(int64) :: clk0, clk_prev, clk_now, clk_rate

  nargs = command_argument_count()
  if (nargs >= 1) then
    call get_command_argument(1, nml_file)
  else
    nml_file = "./fort.41"
  end if
  call system_clock(clk0, clk_rate)
  clk_prev = clk0
  print *, "- READ SETUP NAMELIST"
  call read_setup_namelist(trim(nml_file))
  call setup_ranks()
  if (nranks > 1) print '(a,i0,a,i0)', " - DRIVER IMAGE ", myrank, " OF ", nranks
  f32_out = is_nc_name(output_file)
  call get_environment_variable("MPASSIT_HOST_ARRAYS", envbuf)
  ! (variables of classic files are byte ranges; a NetCDF-4 file at either end -- chunked, deflated, little-endian -- goes through libhdf5 and host arrays)
  dev_flow = f32_out .and. len_trim(envbuf) == 0 .and. nc_is_classic(grid_file_input_grid) .and. &
             (.not. interp_hist .or. nc_is_classic(hist_file_input_grid)) .and. (.not. interp_diag .or. nc_is_classic(diag_file_input_grid))
  if (f32_out) dev_flow = dev_flow .and. nc_output_format() /= 4
  ! several images share one output file through its variables' byte ranges and regrid row blocks of device-resident fields: that is the
  ! device flow only -- said here, before any image has sized an array for a flow that does not exist
  if (nranks > 1 .and. .not. dev_flow) call fatal("several driver images need NetCDF CLASSIC files in and out (the device-resident flow; "// &
                                                  "NetCDF-4, the raw container and MPASSIT_HOST_ARRAYS go through one image)", nranks)
  if (dev_flow) print *, "- NETCDF IN AND OUT: FIELDS STAY ON THE DEVICE BETWEEN THE FILES"
  ! one image per GPU: MPASSIT_DEVICE (tools/mpassit_ranks.py sets it), else the rank's number on its node modulo the node's GPUs
  ! (a rank started by mpiexec / srun), else device 0
  call get_environment_variable("MPASSIT_DEVICE", envbuf)
  gpu = 0
  if (len_trim(envbuf) > 0) then
    read (envbuf, *) gpu
  else if (local_rank > 0) then
    call mpg_check(mpg_device_count(ngpu), "COUNTING GPUS")
    if (ngpu > 0) gpu = mod(local_rank, int(ngpu))
  end if
  call get_environment_variable("MPASSIT_NO_RESERVE", envbuf)   ! A/B switches for measurements (tools/config4_file_job.py)
  no_reserve = len_trim(envbuf) > 0
  call get_environment_variable("MPASSIT_NO_WINDOW", envbuf)
  no_window = len_trim(envbuf) > 0
  if (dev_flow .and. myrank == 0 .and. .not. no_reserve) call reserve_output()   ! the output file's pages are allocated while the inputs are read
  call mpg_check(mpg_init(int(gpu, c_int)), "INITIALIZING GPU RUNTIME")
  call mpg_apply_tune_env()      ! MPASSIT_TUNE: a site's run-time choices (bilinear line type, fan apex, inside tolerance ...)
  call lap("SETUP + GPU RUNTIME")
  print *, "- DEFINE TARGET GRID"
  call define_target_grid()
  call lap("DEFINE TARGET GRID")
  print *, "- DEFINE INPUT GRID"
  call define_input_grid()
  call lap("DEFINE INPUT GRID")
  if (dev_flow .and. .not. no_window) then
    call plan_source_window()
    call lap("PLAN WEIGHT SETS + SOURCE WINDOW")
  else if (dev_flow) then
    call nc_upload_hgt()
  end if
  print *, "- READ INPUT DATA"
  call read_input_data()
  call lap("READ INPUT DATA")
  print *, "- INTERPOLATE DATA"
  call interp_data()
  call lap("INTERPOLATE DATA")
  print *, "- WRITE DATA"
  call write_to_file()
  call lap("WRITE DATA")
  call mpg_check(mpg_mesh_destroy(input_grid), "IN MeshDestroy")
  call mpg_check(mpg_grid_destroy(target_grid_h), "IN GridDestroy")
  call mpg_check(mpg_finalize(), "IN Finalize")
  call system_clock(clk_now)
  print '(a,f9.3,a)', " - DONE.  TOTAL ", real(clk_now - clk0, dp)/real(clk_rate, dp), " s"

contains

  !> wall seconds of the phase that just ended (the reference prints the phase names only, mpassit.F90:105-137)
  subroutine lap(what)
    character(len=*), intent(in) :: what
    call system_clock(clk_now)
    print '(a,a,a,f9.3,a)', "   [", what, "] ", real(clk_now - clk_prev, dp)/real(clk_rate, dp), " s"
    clk_prev = clk_now
  end subroutine lap

  !> Upper bound of the output file's size from the variable lists, the level counts of the input files and the target
  !! grid, and the start of its allocation on a helper thread (ncio_reserve_start): writing 9 GB into a file that does not
  !! exist yet is page allocation, 1.6-2.9 s of the 2.0-2.4 s the writer spent (profiles/r03_shm_probe.txt); into pages
  !! that exist it is 1.1 s.  A failure here is not an error: the writer then creates the file itself, as before.
  subroutine reserve_output()
    character(len=50), allocatable :: names(:), targets(:)
    type(c_ptr) :: nf
    integer(c_int64_t) :: nz, nso, n2, n3, ns, plane
    integer :: n, i
    if (target_from_file .or. i_target <= 0 .or. j_target <= 0) return     ! grid size not known yet
    n2 = 16; n3 = 5; ns = 0; nz = 0; nso = 0                            ! grid / time variables; MU, PB, Z_C, PH, P
    if (interp_diag) then
      call read_varlist('diaglist', n, names, targets)
      do i = 1, n
        if (trim(names(i)) == 'refl10cm') then
          n3 = n3 + 1
        else
          n2 = n2 + 1
        end if
      end do
      if (ncio_open(diag_file_input_grid, nf) == 0) then
        if (ncio_inq_dim(nf, "nVertLevels", nz) /= 0) nz = 0
        if (ncio_close(nf) /= 0) nz = nz
      end if
    end if
    if (interp_hist) then
      call read_varlist('histlist_2d', n, names, targets); n2 = n2 + n + 1      ! + HGT
      call read_varlist('histlist_3d', n, names, targets); n3 = n3 + n
      call read_varlist('histlist_soil', n, names, targets); ns = n
      if (ncio_open(hist_file_input_grid, nf) == 0) then
        if (ncio_inq_dim(nf, "nVertLevels", nz) /= 0) nz = 0
        if (ncio_inq_dim(nf, "nSoilLevels", nso) /= 0) nso = 0
        if (ncio_close(nf) /= 0) nz = nz
      end if
    end if
    if (nz <= 0) return
    plane = int(i_target + 1, c_int64_t)*int(j_target + 1, c_int64_t)*4
    if (ncio_reserve_start(output_file, (n3*(nz + 1) + ns*max(nso, 1_c_int64_t) + n2)*plane + 1048576) /= 0) &
      print *, "- (output space not reserved ahead of time)"
  end subroutine reserve_output

  !> Device flow: build the weight sets this run will use BEFORE the inputs are read (they stay in the Store cache: the
  !! Stores of interp_data find them there), ask each which source ids it references, and declare the union as the
  !! mesh's source window -- then only that range of every variable is read and held in HBM.  With one image the window
  !! is the part of the mesh under the target grid (configuration 4: 2.48 M of 3.0 M cells); with N images each window is
  !! what that image's row block references, so the bytes read per image fall with N (the reference: every rank reads
  !! everything, input_data.F90:645).
  subroutine plan_source_window()
    character(len=50), allocatable :: names(:), targets(:)
    logical :: need_nstd, need_cons, need_node
    integer :: n, i
    integer(c_int64_t) :: lo, hi
    need_nstd = .false.; need_cons = .false.; need_node = .false.
    if (interp_hist) then
      call read_varlist('histlist_2d', n, names, targets)
      do i = 1, n
        if (any(trim(names(i)) == [character(len=50) :: 'ivgtyp', 'isltyp', 'xland', 'landmask'])) need_nstd = .true.
        if (any(trim(names(i)) == [character(len=50) :: 'snow', 'snowh'])) need_cons = .true.
      end do
      call read_varlist('histlist_3d', n, names, targets)
      do i = 1, n
        if (trim(names(i)) == 'vorticity') need_node = .true.
      end do
    end if
    lo = huge(lo); hi = 0
    call plan_one(MPG_MESHLOC_ELEMENT, MPG_REGRIDMETHOD_BILINEAR, lo, hi)
    if (need_cons) call plan_one(MPG_MESHLOC_ELEMENT, MPG_REGRIDMETHOD_CONSERVE, lo, hi)
    if (need_nstd) call plan_one(MPG_MESHLOC_ELEMENT, MPG_REGRIDMETHOD_NEAREST_STOD, lo, hi)
    if (hi <= lo) then
      lo = 0; hi = 0
    end if
    win0_cell = lo; winn_cell = hi - lo
    call mpg_check(mpg_mesh_set_source_window(input_grid, MPG_MESHLOC_ELEMENT, win0_cell, winn_cell), "IN MeshSetSourceWindow")
    if (need_node) then
      lo = huge(lo); hi = 0
      call plan_one(MPG_MESHLOC_NODE, MPG_REGRIDMETHOD_BILINEAR, lo, hi)
      if (hi <= lo) then
        lo = 0; hi = 0
      end if
      win0_vert = lo; winn_vert = hi - lo
      call mpg_check(mpg_mesh_set_source_window(input_grid, MPG_MESHLOC_NODE, win0_vert, winn_vert), "IN MeshSetSourceWindow")
    end if
    print '(a,i0,a,i0,a,i0,a)', " - SOURCE WINDOW: CELLS ", win0_cell + 1, " .. ", win0_cell + winn_cell, " OF ", nCells_input, " ARE READ"
    call nc_upload_hgt()
  end subroutine plan_source_window

  !> one weight set of the plan: Store, its source range joined to [lo, hi), Release (it stays parked in the cache)
  subroutine plan_one(loc, method, lo, hi)
    integer(c_int), intent(in) :: loc, method
    integer(c_int64_t), intent(inout) :: lo, hi
    integer(c_int64_t) :: a, b
    type(c_ptr) :: rh
    call mpg_check(mpg_regrid_store(input_grid, loc, target_grid_h, MPG_STAGGERLOC_CENTER, method, rh), "IN FieldBundleRegridStore")
    call mpg_check(mpg_handle_source_range(rh, a, b), "IN HandleSourceRange")
    call mpg_check(mpg_handle_release(rh), "IN FieldRegridRelease")
    if (b > a) then
      lo = min(lo, a); hi = max(hi, b)
    end if
  end subroutine plan_one

  subroutine define_target_grid()
    if (target_from_file) then
      call define_target_grid_file(target_grid_h)
    else
      call define_target_grid_params(target_grid_h)
    end if
  end subroutine define_target_grid

  subroutine read_f64(u, name, arr, dims, required)
    integer, intent(in) :: u
    character(len=*), intent(in) :: name
    real(dp), allocatable, intent(out) :: arr(:)
    integer(int64), intent(out) :: dims(3)
    logical, intent(in) :: required
    integer(int32) :: dtype, ndim
    logical :: found
    integer :: ios
    call raw_seek(u, name, dtype, ndim, dims, found)
    if (.not. found) then
      if (required) call fatal("reading field id - "//trim(name), -1)
      return
    end if
    if (dtype /= 0) call fatal("field "//trim(name)//" is not float64", -1)
    allocate (arr(dims(1)*dims(2)*dims(3)))
    read (u, iostat=ios) arr
    if (ios /= 0) call fatal("reading field "//trim(name), ios)
  end subroutine read_f64

  subroutine define_input_grid()
    integer :: u, ios
    integer(int32) :: dtype, ndim
    integer(int64) :: dims(3)
    logical :: found
    real(dp), allocatable :: latCell(:), lonCell(:), latVertex(:), lonVertex(:)
    integer(c_int32_t), allocatable :: voc(:)
    if (nc_is_netcdf(grid_file_input_grid)) then
      call nc_read_grid(grid_file_input_grid, latCell, lonCell, latVertex, lonVertex, voc)
      print *, "- NUMBER OF CELLS ON INPUT GRID ", nCells_input
      print *, "- CREATE MESH -"
      call create_mesh(latCell, lonCell, latVertex, lonVertex, voc)
      return
    end if
    call raw_open_read(grid_file_input_grid, u)
    call read_f64(u, "latCell", latCell, dims, .true.)
    nCells_input = int(dims(1))
    call read_f64(u, "lonCell", lonCell, dims, .true.)
    call read_f64(u, "latVertex", latVertex, dims, .true.)
    nVert_input = int(dims(1))
    call read_f64(u, "lonVertex", lonVertex, dims, .true.)
    call raw_seek(u, "verticesOnCell", dtype, ndim, dims, found)
    if (.not. found .or. dtype /= 1) call fatal("reading verticesOnCell", -1)
    maxEdges_input = int(dims(1))
    allocate (voc(dims(1)*dims(2)))
    read (u, iostat=ios) voc
    if (ios /= 0) call fatal("reading verticesOnCell", ios)
    call read_f64(u, "ter", hgt%src, dims, .true.)
    hgt%name = "ter"; hgt%tname = "HGT"; hgt%nlev = 1
    close (u)
    print *, "- NUMBER OF CELLS ON INPUT GRID ", nCells_input
    print *, "- NUMBER OF NODES ON INPUT GRID ", nVert_input
    print *, "- CREATE MESH -"
    call create_mesh(latCell, lonCell, latVertex, lonVertex, voc)
  end subroutine define_input_grid

  !> ESMF_MeshCreate (model_grid.F90:488-497).  One image of several holds only the part of the mesh its rows of the target
  !! grid can see (mpg_mesh_create_window; the reference hands every PET 1/N of the cells, model_grid.F90:423-438): the same
  !! weights, a geometry ingest and Stores that shrink with the row block.  MPASSIT_WHOLE_MESH=1 keeps the whole mesh (A/B).
  subroutine create_mesh(latCell, lonCell, latVertex, lonVertex, voc)
    real(dp), intent(in) :: latCell(:), lonCell(:), latVertex(:), lonVertex(:)
    integer(c_int32_t), intent(in) :: voc(:)
    character(len=16) :: buf
    integer(c_int64_t) :: c0, cn, v0, vn
    real(c_double) :: margin
    integer, allocatable :: my_cells(:)
    integer :: my_cells_num
    ! block_decomp_file (program_setup.F90:38; model_grid.F90:423-438): the reference gives every PET the cells the MPAS graph
    ! partition assigns to it.  The images of this driver hold what their target ROWS reference instead (source windows, no exchange),
    ! so the file decides nothing here -- but it is read and checked as the reference checks it (exists, lists exactly nCells cells,
    ! was made for exactly this many processes), and the cells it gives this image are what a host that exchanges its sources hands
    ! to mpg_halo_build_owned.
    if (trim(block_decomp_file) /= "NULL") then
      call read_block_decomp_file(myrank, nranks, block_decomp_file, nCells_input, my_cells, my_cells_num)
      print '(a,i0,a,i0,a)', " - BLOCK DECOMPOSITION FILE: ", my_cells_num, " OF ", nCells_input, " CELLS BELONG TO THIS IMAGE (not used: images read by target rows)"
    end if
    call get_environment_variable("MPASSIT_WHOLE_MESH", buf)
    if (nranks > 1 .and. len_trim(buf) == 0) then
      call mpg_check(mpg_mesh_create_window(int(nCells_input, c_int64_t), int(nVert_input, c_int64_t), int(maxEdges_input, c_int), &
                                            latCell, lonCell, latVertex, lonVertex, voc, target_grid_h, input_grid), "IN MeshCreate")
      call mpg_check(mpg_mesh_window_info(input_grid, c0, cn, v0, vn, margin), "IN MeshWindowInfo")
      print '(a,i0,a,i0,a,i0,a)', " - MESH WINDOW OF THIS IMAGE: ", cn, " OF ", nCells_input, " CELLS, ", vn, " VERTICES"
    else
      call mpg_check(mpg_mesh_create(int(nCells_input, c_int64_t), int(nVert_input, c_int64_t), int(maxEdges_input, c_int), &
                                     latCell, lonCell, latVertex, lonVertex, voc, input_grid), "IN MeshCreate")
    end if
  end subroutine create_mesh

  subroutine load_field(u, name, tname, f)
    integer, intent(in) :: u
    character(len=*), intent(in) :: name, tname
    type(field_t), intent(out) :: f
    integer(int64) :: dims(3)
    if (nc_in) then
      call nc_load_field(nf_in, name, tname, f)
      return
    end if
    f%name = name; f%tname = tname
    call read_f64(u, name, f%src, dims, .true.)
    if (dims(2) == 1) then
      f%nlev = 1                 ! [nCells]
    else
      f%nlev = int(dims(1))      ! [nCells][nlev] in the file == Fortran (nlev, nCells)
    end if
  end subroutine load_field

  !> the bundle takes the field's arrays over (no copies: a 3-D source of configuration 4 is 1.3 GB)
  subroutine append(b, f)
    type(bundle_t), intent(inout) :: b
    type(field_t), intent(inout) :: f
    type(field_t), allocatable :: tmp(:)
    integer :: i
    allocate (tmp(b%n + 1))
    do i = 1, b%n
      call move_field(b%f(i), tmp(i))
    end do
    call move_field(f, tmp(b%n + 1))
    call move_alloc(tmp, b%f)
    b%n = b%n + 1
  end subroutine append

  subroutine read_input_data()
    character(len=50), allocatable :: names(:), targets(:)
    character(len=50) :: cons_vars(2), nstd_vars(4), nzp1_vars(2)
    type(field_t) :: f
    integer :: n, i, u
    cons_vars = [character(len=50) :: 'snow', 'snowh']                       ! input_data.F90:840-842
    nstd_vars = [character(len=50) :: 'ivgtyp', 'isltyp', 'xland', 'landmask']
    nzp1_vars = [character(len=50) :: 'zgrid', 'w']
    if (interp_diag) then
      call read_varlist('diaglist', n, names, targets)
      call open_in(diag_file_input_grid, u, .true.)
      do i = 1, n
        call load_field(u, names(i), targets(i), f)
        call append(diag_bundle, f)
        if (trim(names(i)) == 'u10') u10_ind = i
        if (trim(names(i)) == 'v10') v10_ind = i
      end do
      call close_in(u)
    end if
    if (interp_hist) then
      call open_in(hist_file_input_grid, u, .false.)
      call read_varlist('histlist_2d', n, names, targets)
      do i = 1, n
        call load_field(u, names(i), targets(i), f)
        if (is_in(names(i), cons_vars)) then
          call append(hist_2d_cons, f)
        else if (is_in(names(i), nstd_vars)) then
          call append(hist_2d_nstd, f)
        else
          call append(hist_2d_patch, f)
        end if
      end do
      call read_varlist('histlist_3d', n, names, targets)
      do i = 1, n
        call load_field(u, names(i), targets(i), f)
        if (wrf_mod_vars .and. trim(names(i)) == 'uReconstructZonal') then
          do_u_interp = 1; call move_field(f, u_field); umass%name = 'UMASS'; umass%tname = 'UMASS'
        else if (wrf_mod_vars .and. trim(names(i)) == 'uReconstructMeridional') then
          do_v_interp = 1; call move_field(f, v_field); vmass%name = 'VMASS'; vmass%tname = 'VMASS'
        else if (is_in(names(i), nzp1_vars)) then
          call append(hist_3d_nzp1, f)
        else if (trim(names(i)) == 'vorticity') then
          call append(hist_3d_vert, f)
        else
          call append(hist_3d_nz, f)
        end if
      end do
      call read_varlist('histlist_soil', n, names, targets)
      do i = 1, n
        call load_field(u, names(i), targets(i), f)
        call append(hist_soil, f)
      end do
      call close_in(u)
    end if
  end subroutine read_input_data

  subroutine open_in(file, u, is_diag)
    character(len=*), intent(in) :: file
    integer, intent(out) :: u
    logical, intent(in) :: is_diag
    integer(c_int) :: id, rc
    integer(c_int8_t) :: tb(64)
    nc_in = nc_is_netcdf(file)
    u = -1
    if (nc_in) then
      nc_in_path = file
      call ncio_check(ncio_open(file, nf_in), "opening "//trim(file))
      call nc_read_meta(nf_in, is_diag)
      if (ncio_inq_varid(nf_in, "xtime", id) == 0) then
        tb = 32_c_int8_t
        rc = ncio_get_var(nf_in, id, 0_c_int64_t, NCIO_CHAR, tb)
        if (rc == 0) valid_time = transfer(tb(1:19), valid_time)
      end if
    else
      call raw_open_read(file, u)
    end if
  end subroutine open_in

  subroutine close_in(u)
    integer, intent(in) :: u
    if (nc_in) then
      call ncio_check(ncio_close(nf_in), "closing input file")
      nc_in = .false.
    else
      close (u)
    end if
  end subroutine close_in

  subroutine put(u, f, ni, nj)
    integer, intent(in) :: u, ni, nj
    type(field_t), intent(in) :: f
    integer(int64) :: dims(3)
    if (.not. allocated(f%dst)) return
    dims = [int(ni, int64), int(nj, int64), int(f%nlev, int64)]
    call raw_write_f64(u, trim(f%tname), merge(2, 3, f%nlev == 1), dims, f%dst)
  end subroutine put

  subroutine put_bundle(u, b)
    integer, intent(in) :: u
    type(bundle_t), intent(in) :: b
    integer :: i
    do i = 1, b%n
      call put(u, b%f(i), i_target, j_target)
    end do
  end subroutine put_bundle

  logical function is_nc_name(file)
    character(len=*), intent(in) :: file
    integer :: l
    l = len_trim(file)
    is_nc_name = .false.
    if (l > 3) is_nc_name = file(l - 2:l) == ".nc"
  end function is_nc_name

  subroutine write_to_file()
    integer :: u
    if (is_nc_name(output_file)) then
      call nc_write_target(trim(output_file), valid_time)
      return
    end if
    call raw_open_write(output_file, u)
    if (interp_diag) call put_bundle(u, diag_bundle)
    if (interp_hist) then
      call put(u, hgt, i_target, j_target)
      call put_bundle(u, hist_2d_patch)
      call put_bundle(u, hist_2d_cons)
      call put_bundle(u, hist_2d_nstd)
      call put_bundle(u, hist_3d_nz)
      call put_bundle(u, hist_3d_nzp1)
      call put_bundle(u, hist_3d_vert)
      call put_bundle(u, hist_soil)
      if (do_u_interp == 1) then
        call put(u, umass, i_target, j_target)
        call put(u, u_field, i_target + 1, j_target)
      end if
      if (do_v_interp == 1) then
        call put(u, vmass, i_target, j_target)
        call put(u, v_field, i_target, j_target + 1)
      end if
    end if
    close (u)
  end subroutine write_to_file

end program mpassit
