!> Host-side support modules of the Fortran driver: namelist, variable lists, target grid and a raw-binary array
!! container (the first file surface of this build; NetCDF classic files go through ncio_mod / ncfiles_mod).
!!   program_setup  <- program_setup.F90:87-249 (namelist &config :103-106, derived sizes :163-164,:238-240)
!!   varlists       <- input_data.F90:1146-1194 (read_varlist), :840-911 (classification)
!!   target_grid    <- model_grid.F90:644-1201: 'lambert' / 'lat-lon' grids are evaluated on the GPU from the projection
!!                     scalars (mpg_grid_create_proj replaces get_lat_lon_fields :2188-2219, get_rotang :2450-2507 and
!!                     get_map_factor :2229-2365); 'file' grids are read from a WRF-style file (:1203-1888) and their
!!                     corners rebuilt by get_cell_corners (:1902-1972)
!! Reals are real(8) explicitly (the reference promotes with -r8 / -fdefault-real-8, CMakeLists.txt:80-82).

module program_setup
  implicit none
  public
  integer, parameter :: dp = kind(1.0d0)
  real(dp), parameter :: NAN = 1.0e20_dp           ! misc_definitions_module.F90:12 ("unset" sentinel)
  integer, parameter :: PROJ_LATLON = 0, PROJ_LC = 1, PROJ_PS = 2, PROJ_MERC = 3  ! misc_definitions_module.F90:38-42
  real(dp), parameter :: PI = 3.141592653589793_dp, RAD_PER_DEG = PI/180.0_dp, DEG_PER_RAD = 180.0_dp/PI
  real(dp), parameter :: EARTH_RADIUS_M = 6370000.0_dp

  character(len=500) :: grid_file_input_grid = "NULL", diag_file_input_grid = "NULL", hist_file_input_grid = "NULL"
  character(len=500) :: file_target_grid = "NULL", output_file = "NULL", block_decomp_file = "NULL"
  character(len=500) :: target_grid_type = "lambert"
  logical :: interp_diag = .false., interp_hist = .false., wrf_mod_vars = .false., is_regional = .true.
  logical :: interp_as_bundle = .true., esmf_log = .false.
  logical :: target_from_file = .false.            ! target_grid_type = 'file'
  integer :: i_target = 0, j_target = 0, proj_code = PROJ_LC
  real(dp) :: truelat1 = NAN, truelat2 = NAN, stand_lon = NAN, ref_lat = NAN, ref_lon = NAN, ref_x = NAN, ref_y = NAN
  real(dp) :: pole_lat = 90.0_dp, pole_lon = 0.0_dp
  character(len=500) :: map_proj_char = ""          ! program_setup.F90:172-187, model_grid.F90:1288-1295
  real(dp) :: dxkm, dykm, dlondeg, dlatdeg, known_lat, known_lon, known_x, known_y
  ! One driver image per GPU (the reference's PETs, mpassit.F90:84-96): image `myrank` of `nranks` owns the target mass rows
  ! j_lo..j_hi (the split of regDecomp=(/1,npets/), model_grid.F90:693, by para_range :2428-2441) and regrids rows
  ! je_lo..je_hi -- its own plus one halo row each side, which the CENTER -> EDGE destaggering of its first / last row needs.
  ! Every image reads the input files whole, like every rank of the reference (input_data.F90:645); no data moves between
  ! the images.  Set by the launcher: MPASSIT_NRANKS / MPASSIT_RANK / MPASSIT_RUN_ID (tools/mpassit_ranks.py), or the variables of mpiexec / srun
  ! (setup_ranks below).
  integer :: nranks = 1, myrank = 0, local_rank = 0, j_lo = 1, j_hi = 0, je_lo = 1, je_hi = 0, ny_ext = 0
  character(len=64) :: run_id = "0"

contains

  !> model_grid.F90:2428-2441: inclusive 1-based block [ista, iend] of rank irank
  subroutine para_range(n1, n2, nprocs, irank, ista, iend)
    integer, intent(in) :: n1, n2, nprocs, irank
    integer, intent(out) :: ista, iend
    integer :: iwork1, iwork2
    iwork1 = (n2 - n1 + 1)/nprocs
    iwork2 = mod(n2 - n1 + 1, nprocs)
    ista = irank*iwork1 + n1 + min(irank, iwork2)
    iend = ista + iwork1 - 1
    if (iwork2 > irank) iend = iend + 1
  end subroutine para_range

  !> read_block_decomp_file (model_grid.F90:2367-2426): the namelist's block_decomp_file is an MPAS graph partition file, one line per
  !! cell with the PET that owns it (blank lines skipped).  Checks as the reference: the file exists, lists exactly ncells cells and was
  !! made for exactly npets processes.  my_cells: this image's cells, sorted, 0-BASED -- the owned_ids of mpg_halo_build_owned.
  subroutine read_block_decomp_file(localpet, npets, file, ncells, my_cells, my_cells_num)
    integer, intent(in) :: localpet, npets, ncells
    character(len=*), intent(in) :: file
    integer, allocatable, intent(out) :: my_cells(:)      ! (default integers: 32 bits, what the C side takes as int32_t)
    integer, intent(out) :: my_cells_num
    logical :: ex
    integer :: u, k, istat, nlines, proc, proc_max
    integer, allocatable :: tmp(:)
    character(len=200) :: line
    character(len=200) :: msg
    inquire (file=trim(file), exist=ex)
    if (.not. ex) call fatal("BLOCK DECOMP FILE DOES NOT EXIST", -1)
    open (newunit=u, file=trim(file), form='formatted', status='old', iostat=istat)
    if (istat /= 0) call fatal("OPENING BLOCK DECOMP FILE", istat)
    nlines = 0
    do
      read (u, '(A)', iostat=istat) line
      if (istat /= 0) exit
      if (trim(line) == '') cycle
      nlines = nlines + 1
    end do
    if (nlines /= ncells) call fatal("BLOCK DECOMPOSITION FILE CONTAINS MORE CELLS THAN INPUT GRID", -1)
    allocate (tmp(ncells))
    my_cells_num = 0
    proc_max = 0
    rewind (u)
    k = 0
    do while (k < nlines)
      read (u, '(A)', iostat=istat) line
      if (istat /= 0) call fatal("READING BLOCK DECOMPOSITION FILE", istat)
      if (trim(line) == '') cycle
      read (line, *, iostat=istat) proc
      if (istat /= 0) call fatal("READING BLOCK DECOMPOSITION FILE", istat)
      k = k + 1
      proc_max = max(proc, proc_max)
      if (localpet == proc) then
        my_cells_num = my_cells_num + 1
        tmp(my_cells_num) = k - 1
      end if
    end do
    close (u)
    if (proc_max + 1 /= npets) then
      write (msg, '(A,I10,A,I10,A)') "BLOCK DECOMPOSITION FILE GENERATED FOR ", proc_max + 1, " PROCESSES BUT ", npets, " PROCESSORS USED."
      call fatal(trim(msg), -1)
    end if
    allocate (my_cells(my_cells_num))
    my_cells(1:my_cells_num) = tmp(1:my_cells_num)
  end subroutine read_block_decomp_file

  !> Who am I among how many driver images?  In this order:
  !!   1. MPASSIT_NRANKS / MPASSIT_RANK / MPASSIT_RUN_ID        -- tools/mpassit_ranks.py (no MPI anywhere)
  !!   2. the variables an MPI or Slurm launcher gives its ranks   -- `mpiexec -n 8 mpassit namelist.input`, `srun -n 8 ...`, the
  !!      reference's own launch lines (mpassit.F90:84-96 asks MPI; this driver links no MPI and only reads the launcher's environment):
  !!        MPICH / hydra   PMI_SIZE, PMI_RANK, MPI_LOCALRANKID;        run tag: the launcher's proxy (parent process id)
  !!        Open MPI        OMPI_COMM_WORLD_SIZE, _RANK, _LOCAL_RANK;   run tag: PMIX_NAMESPACE (or the parent process id)
  !!        Slurm srun      SLURM_NTASKS, SLURM_PROCID, SLURM_LOCALID;  run tag: SLURM_JOB_ID.SLURM_STEP_ID -- only inside a job STEP
  !!                        (SLURM_STEP_ID set): a plain `./mpassit` in a batch script also sees SLURM_NTASKS and must stay one image
  !!   3. one image.
  !! local_rank is the rank's number on its node: the GPU it takes when MPASSIT_DEVICE says nothing (modulo the node's GPUs).
  subroutine setup_ranks()
    character(len=64) :: buf
    integer :: ios
    interface
      function c_getppid() bind(C, name="getppid") result(p)
        use, intrinsic :: iso_c_binding, only: c_int
        integer(c_int) :: p
      end function c_getppid
    end interface
    local_rank = 0
    call get_environment_variable("MPASSIT_NRANKS", buf)
    if (len_trim(buf) > 0) then
      read (buf, *, iostat=ios) nranks
      if (ios /= 0 .or. nranks < 1) call fatal("MPASSIT_NRANKS must be a positive integer", ios)
      call get_environment_variable("MPASSIT_RANK", buf)
      if (len_trim(buf) > 0) then
        read (buf, *, iostat=ios) myrank
        if (ios /= 0 .or. myrank < 0 .or. myrank >= nranks) call fatal("MPASSIT_RANK must lie in 0 .. MPASSIT_NRANKS-1", ios)
      end if
      local_rank = myrank
      call get_environment_variable("MPASSIT_RUN_ID", buf)
      if (len_trim(buf) > 0) run_id = buf
      ! the images find each other through marker files named after the run id: a constant default would let the markers of
      ! a killed earlier run release this one's images early (tools/mpassit_ranks.py draws a fresh id per launch)
      if (nranks > 1 .and. len_trim(buf) == 0) call fatal("MPASSIT_NRANKS > 1 needs MPASSIT_RUN_ID, unique per launch", nranks)
      return
    end if
    if (from_launcher("PMI_SIZE", "PMI_RANK", "MPI_LOCALRANKID")) then
      write (run_id, '(a,i0)') "hydra", c_getppid()
      call node_local_tag_check("MPI_LOCALNRANKS")
    else if (from_launcher("OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_LOCAL_RANK")) then
      call get_environment_variable("PMIX_NAMESPACE", buf)
      if (len_trim(buf) > 0) then
        run_id = "ompi"//tag_of(buf)
      else
        write (run_id, '(a,i0)') "ompi", c_getppid()
        call node_local_tag_check("OMPI_COMM_WORLD_LOCAL_SIZE")
      end if
    else
      call get_environment_variable("SLURM_STEP_ID", buf)
      if (len_trim(buf) > 0) then
        if (from_launcher("SLURM_NTASKS", "SLURM_PROCID", "SLURM_LOCALID")) then
          run_id = "slurm"//trim(buf)
          call get_environment_variable("SLURM_JOB_ID", buf)
          run_id = trim(run_id)//"j"//trim(buf)
        end if
      end if
    end if
    ! MPASSIT_RUN_ID is the user's word on what belongs together: it overrides whatever tag the launcher's variables gave
    ! (round-5 advisor: it was ignored on this path, so a multi-node hydra launch could not be given a job-wide tag at all)
    call get_environment_variable("MPASSIT_RUN_ID", buf)
    if (nranks > 1 .and. len_trim(buf) > 0) run_id = "user"//tag_of(buf)
    if (nranks > 1) print '(a,i0,a,i0,a,a)', " - LAUNCHED AS RANK ", myrank, " OF ", nranks, " BY AN MPI / SLURM LAUNCHER; RUN TAG ", trim(run_id)
  contains
    !> The parent process id is the launcher's PER-NODE proxy (hydra_pmi_proxy, orted): images on different nodes would get different
    !! tags, look for different marker files and wait MPASSIT_WAIT_S for images that are there all along.  When the launcher says that
    !! fewer ranks share this node than the job has (v_local_size), a job-wide tag is needed: MPASSIT_RUN_ID, or stop now saying so.
    subroutine node_local_tag_check(v_local_size)
      character(len=*), intent(in) :: v_local_size
      character(len=64) :: b, id
      integer :: nloc, e
      call get_environment_variable(v_local_size, b)
      if (len_trim(b) == 0) return
      read (b, *, iostat=e) nloc
      if (e /= 0 .or. nloc < 1 .or. nloc >= nranks) return
      call get_environment_variable("MPASSIT_RUN_ID", id)
      if (len_trim(id) > 0) return
      call fatal("this launch spans several nodes ("//trim(b)//" of its ranks on this one) and the launcher gives no job-wide name: "// &
                 "set MPASSIT_RUN_ID to one value, unique per launch, for all ranks (e.g. mpiexec -genv MPASSIT_RUN_ID $$)", nranks)
    end subroutine node_local_tag_check

    !> nranks / myrank / local_rank from a launcher's three variables; .false. (and nothing changed) unless the first two are there and sane
    logical function from_launcher(v_size, v_rank, v_local)
      character(len=*), intent(in) :: v_size, v_rank, v_local
      character(len=64) :: b
      integer :: n, r, l, e1, e2
      from_launcher = .false.
      call get_environment_variable(v_size, b)
      if (len_trim(b) == 0) return
      read (b, *, iostat=e1) n
      call get_environment_variable(v_rank, b)
      if (len_trim(b) == 0) return
      read (b, *, iostat=e2) r
      if (e1 /= 0 .or. e2 /= 0 .or. n < 1 .or. r < 0 .or. r >= n) return
      nranks = n
      myrank = r
      local_rank = r
      call get_environment_variable(v_local, b)
      if (len_trim(b) > 0) then
        read (b, *, iostat=e1) l
        if (e1 == 0 .and. l >= 0) local_rank = l
      end if
      from_launcher = .true.
    end function from_launcher
    !> letters and digits of a launcher's job name (it goes into file names)
    function tag_of(txt) result(t)
      character(len=*), intent(in) :: txt
      character(len=40) :: t
      integer :: i, k
      t = ""
      k = 0
      do i = 1, len_trim(txt)
        if (k >= 40) exit
        if (verify(txt(i:i), "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789") == 0) then
          k = k + 1
          t(k:k) = txt(i:i)
        end if
      end do
    end function tag_of
  end subroutine setup_ranks

  !> row blocks once the target grid's size is known
  subroutine setup_row_block()
    if (nranks > j_target) call fatal("more driver images than target rows", nranks)
    call para_range(1, j_target, nranks, myrank, j_lo, j_hi)
    je_lo = max(j_lo - 1, 1)
    je_hi = min(j_hi + 1, j_target)
    if (nranks == 1) then
      je_lo = 1; je_hi = j_target
    end if
    ny_ext = je_hi - je_lo + 1
  end subroutine setup_row_block

  subroutine fatal(msg, code)
    character(len=*), intent(in) :: msg
    integer, intent(in) :: code
    print *, "- FATAL ERROR: "
    write (*, '(A)') trim(msg)
    print *, "- IOSTAT IS: ", code
    error stop 999
  end subroutine fatal

  function upper(s) result(u)
    character(len=*), intent(in) :: s
    character(len=len(s)) :: u
    integer :: i, c
    u = s
    do i = 1, len(s)
      c = iachar(s(i:i))
      if (c >= iachar('a') .and. c <= iachar('z')) u(i:i) = achar(c - 32)
    end do
  end function upper

  subroutine read_setup_namelist(filename)
    character(len=*), intent(in) :: filename
    real(dp) :: dx, dy
    integer :: nx, ny, ierr, u
    character(len=500) :: kind
    namelist /config/ grid_file_input_grid, diag_file_input_grid, hist_file_input_grid, file_target_grid, output_file, &
      interp_diag, interp_hist, wrf_mod_vars, esmf_log, target_grid_type, nx, ny, dx, dy, ref_lat, ref_lon, ref_x, ref_y, &
      truelat1, truelat2, stand_lon, is_regional, pole_lat, pole_lon, interp_as_bundle, block_decomp_file
    dx = NAN; dy = NAN; nx = 0; ny = 0
    open (newunit=u, file=trim(filename), status='old', iostat=ierr)
    if (ierr /= 0) call fatal("OPENING SETUP NAMELIST.", ierr)
    read (u, nml=config, iostat=ierr)
    if (ierr /= 0) call fatal("READING SETUP NAMELIST.", ierr)
    close (u)
    kind = upper(trim(target_grid_type))
    if (trim(kind) == 'FILE') then               ! everything else comes from the file (define_target_grid_file)
      if (trim(file_target_grid) == "NULL") call fatal("target_grid_type='file' needs file_target_grid", 3)
      target_from_file = .true.
      return
    end if
    dxkm = dx; dykm = dy
    known_lat = ref_lat; known_lon = ref_lon; known_x = ref_x; known_y = ref_y
    i_target = nx - 1; j_target = ny - 1        ! namelist nx, ny are STAGGERED counts
    if (trim(kind) == 'LAMBERT') then
      proj_code = PROJ_LC
      map_proj_char = 'Lambert Conformal'
      if (truelat2 == NAN) then
        if (truelat1 == NAN) call fatal("No TRUELAT1 specified for Lambert conformal projection.", 3)
        truelat2 = truelat1
      end if
    else if (trim(kind) == 'MERCATOR') then        ! program_setup.F90:174-177
      proj_code = PROJ_MERC
      map_proj_char = 'Mercator'
    else if (trim(kind) == 'POLAR') then           ! :179-182
      proj_code = PROJ_PS
      map_proj_char = 'Polar Stereographic'
    else if (trim(kind) == 'LAT-LON') then
      proj_code = PROJ_LATLON
      map_proj_char = 'Lat/Lon'
      if (dx == NAN .and. dy == NAN) then
        if (is_regional) call fatal("For lat-lon projection, if dx/dy are not specified a global grid is assumed.", 3)
        dlondeg = 360.0_dp/i_target; dlatdeg = 180.0_dp/j_target
        known_x = 1.0_dp; known_y = 1.0_dp
        known_lon = stand_lon + dlondeg/2.0_dp; known_lat = -90.0_dp + dlatdeg/2.0_dp
        dxkm = EARTH_RADIUS_M*PI*2.0_dp/i_target; dykm = EARTH_RADIUS_M*PI/j_target          ! program_setup.F90:209-210
      else
        if (.not. is_regional) call fatal("For lat-lon projection, if dx/dy are specified a regional grid is assumed.", 3)
        dlatdeg = dy; dlondeg = dx
        dxkm = dlondeg*EARTH_RADIUS_M*PI*2.0_dp/360.0_dp; dykm = dlatdeg*EARTH_RADIUS_M*PI*2.0_dp/360.0_dp   ! :221-222
        if (known_lat == NAN .or. known_lon == NAN) call fatal("lat-lon with dx/dy needs ref_lat, ref_lon", 3)
      end if
    else
      call fatal('In namelist, invalid target_grid_type specified. Valid projections are "lambert", "mercator", "polar", '// &
                 '"lat-lon" and "file".', 3)
    end if
    if (known_x == NAN .and. known_y == NAN) then
      known_x = real(i_target + 1, dp)/2.0_dp; known_y = real(j_target + 1, dp)/2.0_dp
    else if (known_x == NAN .or. known_y == NAN) then
      call fatal("In namelist, neither or both of ref_x, ref_y must be specified.", 3)
    end if
  end subroutine read_setup_namelist
end module program_setup

!> Two-column variable lists (fixed names in the CWD: diaglist, histlist_2d, histlist_3d, histlist_soil).
module varlists
  use program_setup, only: fatal
  implicit none
  public
contains
  subroutine read_varlist(file, n, names, targets)
    character(len=*), intent(in) :: file
    integer, intent(out) :: n
    character(len=50), allocatable, intent(out) :: names(:), targets(:)
    integer :: u, istat, k
    character(len=200) :: line
    logical :: ex
    inquire (file=trim(file), exist=ex)
    if (.not. ex) call fatal("VARLIST FILE "//trim(file)//" not exist", 1)
    open (newunit=u, file=trim(file), form='formatted', status='old', iostat=istat)
    if (istat /= 0) call fatal("OPENING VARLIST FILE", istat)
    n = 0
    do
      read (u, '(A)', iostat=istat) line
      if (istat /= 0) exit
      if (trim(line) == '') cycle
      n = n + 1
    end do
    allocate (names(n), targets(n))
    rewind (u)
    k = 0
    do while (k < n)
      read (u, '(A)', iostat=istat) line
      if (istat /= 0) call fatal("READING VARLIST FILE", istat)
      if (trim(line) == '') cycle
      k = k + 1
      read (line, *, iostat=istat) names(k), targets(k)
      if (istat /= 0) call fatal("READING VARLIST FILE", istat)
    end do
    close (u)
  end subroutine read_varlist

  logical function is_in(name, list)
    character(len=*), intent(in) :: name, list(:)
    integer :: i
    is_in = .false.
    do i = 1, size(list)
      if (trim(name) == trim(list(i))) is_in = .true.
    end do
  end function is_in
end module varlists

!> Target grid of a `lambert` / `lat-lon` namelist (define_target_grid_params, model_grid.F90:736-1038): the four
!! staggers, cos/sin(alpha) and the map factors are evaluated on the GPU from the projection scalars
!! (mpg_grid_create_proj); the host keeps copies of what the output file carries (write_data.F90:1003-1140).
module target_grid
  use, intrinsic :: iso_c_binding
  use program_setup
  use mpg
  use ncio
  implicit none
  public
  real(dp), allocatable :: lat_m(:, :), lon_m(:, :), lat_u(:, :), lon_u(:, :), lat_v(:, :), lon_v(:, :)
  real(dp), allocatable, target :: cosa(:, :), sina(:, :), mapfac_m(:, :), mapfac_u(:, :), mapfac_v(:, :)
contains
  subroutine define_target_grid_params(grid_h)
    type(c_ptr), intent(out) :: grid_h
    type(mpg_proj) :: p
    integer(c_int) :: flags
    real(dp), allocatable :: lat_c(:, :), lon_c(:, :)
    p%code = int(proj_code, c_int)                           ! the arguments of map_set (model_grid.F90:676-678)
    p%known_lat = known_lat; p%known_lon = known_lon; p%known_x = known_x; p%known_y = known_y
    p%dx_m = 0.0_dp; p%stand_lon = 0.0_dp; p%truelat1 = 0.0_dp; p%truelat2 = 0.0_dp; p%dlat_deg = 0.0_dp; p%dlon_deg = 0.0_dp
    if (proj_code == PROJ_LC) then
      p%dx_m = dxkm; p%stand_lon = stand_lon; p%truelat1 = truelat1; p%truelat2 = truelat2
    else if (proj_code == PROJ_PS) then                      ! llxy_module.F90:123-132
      p%dx_m = dxkm; p%stand_lon = stand_lon; p%truelat1 = truelat1
    else if (proj_code == PROJ_MERC) then                    ! llxy_module.F90:71-79
      p%dx_m = dxkm; p%truelat1 = truelat1
    else
      p%dlat_deg = dlatdeg; p%dlon_deg = dlondeg
    end if
    flags = 0
    if (.not. is_regional) flags = MPG_GRID_PERIODIC_I       ! ESMF_GridCreate1PeriDim (model_grid.F90:685-694)
    call mpg_check(mpg_grid_create_proj(p, int(i_target, c_int), int(j_target, c_int), flags, grid_h), "IN GridCreate")
    allocate (lat_m(i_target, j_target), lon_m(i_target, j_target), mapfac_m(i_target, j_target))
    allocate (lat_u(i_target + 1, j_target), lon_u(i_target + 1, j_target), mapfac_u(i_target + 1, j_target))
    allocate (lat_v(i_target, j_target + 1), lon_v(i_target, j_target + 1), mapfac_v(i_target, j_target + 1))
    call mpg_check(mpg_grid_get_coords(grid_h, MPG_STAGGERLOC_CENTER, lon_m, lat_m), "IN GridGetCoord")
    call mpg_check(mpg_grid_get_coords(grid_h, MPG_STAGGERLOC_EDGE1, lon_u, lat_u), "IN GridGetCoord")
    call mpg_check(mpg_grid_get_coords(grid_h, MPG_STAGGERLOC_EDGE2, lon_v, lat_v), "IN GridGetCoord")
    call mpg_check(mpg_grid_get_mapfac(grid_h, MPG_STAGGERLOC_CENTER, mapfac_m), "IN get_map_factor")
    call mpg_check(mpg_grid_get_mapfac(grid_h, MPG_STAGGERLOC_EDGE1, mapfac_u), "IN get_map_factor")
    call mpg_check(mpg_grid_get_mapfac(grid_h, MPG_STAGGERLOC_EDGE2, mapfac_v), "IN get_map_factor")
    if (proj_code == PROJ_LC) then                           ! get_rotang only runs for Lambert (model_grid.F90:1113)
      allocate (cosa(i_target, j_target), sina(i_target, j_target))
      call mpg_check(mpg_grid_get_rotang(grid_h, cosa, sina), "IN get_rotang")
    end if
    call setup_row_block()
    if (nranks > 1) then
      allocate (lat_c(i_target + 1, j_target + 1), lon_c(i_target + 1, j_target + 1))
      call mpg_check(mpg_grid_get_coords(grid_h, MPG_STAGGERLOC_CORNER, lon_c, lat_c), "IN GridGetCoord")
      call mpg_check(mpg_grid_destroy(grid_h), "IN GridDestroy")
      call create_row_block_grid(lat_c, lon_c, grid_h)
      ! the block's arrays are rows je_lo .. of this projection's grid: its Stores may search through the inverse projection
      ! (the library checks the claim on the grid's own points)
      call mpg_check(mpg_grid_attach_proj(grid_h, p, int(je_lo - 1, c_int)), "IN GridAttachProj")
    end if
  end subroutine define_target_grid_params

  !> the grid object of this image's row block je_lo..je_hi, from the rows of the full coordinate arrays (the same numbers
  !! the single-image run regrids to, so the results are the same bits); a global grid keeps only the pole caps it touches
  subroutine create_row_block_grid(lat_c, lon_c, grid_h)
    real(dp), intent(in) :: lat_c(:, :), lon_c(:, :)
    type(c_ptr), intent(out) :: grid_h
    integer(c_int) :: flags
    flags = 0
    if (.not. is_regional) then
      flags = MPG_GRID_PERIODIC_I
      if (je_lo > 1) flags = ior(flags, MPG_GRID_NO_SOUTH_POLE)
      if (je_hi < j_target) flags = ior(flags, MPG_GRID_NO_NORTH_POLE)
    end if
    call mpg_check(mpg_grid_create(int(i_target, c_int), int(ny_ext, c_int), flags, lon_m(:, je_lo:je_hi), lat_m(:, je_lo:je_hi), &
                                   lon_c(:, je_lo:je_hi + 1), lat_c(:, je_lo:je_hi + 1), lon_u(:, je_lo:je_hi), lat_u(:, je_lo:je_hi), &
                                   lon_v(:, je_lo:je_hi + 1), lat_v(:, je_lo:je_hi + 1), grid_h), "IN GridCreate (row block)")
  end subroutine create_row_block_grid

  !> target_grid_type = 'file' (define_target_grid_file, model_grid.F90:1203-1888): dimensions, projection attributes,
  !! XLONG|XLONG_M, XLAT|XLAT_M, the U / V staggers, MAPFAC_M/U/V and (Lambert) SINALPHA / COSALPHA come from a WRF
  !! geo_em / wrfinput style file in a NetCDF classic format; the CORNER stagger is rebuilt by get_cell_corners.
  subroutine define_target_grid_file(grid_h)
    type(c_ptr), intent(out) :: grid_h
    type(c_ptr) :: nf
    integer(c_int64_t) :: ni, nj
    real(dp), allocatable :: lat_c(:, :), lon_c(:, :)
    real(dp) :: v
    type(mpg_proj) :: p
    integer(c_int) :: rc_attach
    call ncio_check(ncio_open(trim(file_target_grid), nf), "opening "//trim(file_target_grid))
    call ncio_check(ncio_inq_dim(nf, "west_east", ni), "reading west_east")
    call ncio_check(ncio_inq_dim(nf, "south_north", nj), "reading south_north")
    i_target = int(ni); j_target = int(nj)
    call ncio_check(ncio_get_gatt(nf, "DX", dxkm), "reading DX")
    dykm = dxkm
    proj_code = PROJ_LC
    if (ncio_get_gatt(nf, "MAP_PROJ", v) == 0) proj_code = nint(v)
    if (proj_code < PROJ_LATLON .or. proj_code > PROJ_MERC) call fatal("file_target_grid: unsupported MAP_PROJ", proj_code)
    if (ncio_get_gatt(nf, "STAND_LON", v) == 0) stand_lon = v
    if (ncio_get_gatt(nf, "TRUELAT1", v) == 0) truelat1 = v
    if (ncio_get_gatt(nf, "TRUELAT2", v) == 0) truelat2 = v
    if (ncio_get_gatt(nf, "CEN_LAT", v) == 0) ref_lat = v
    if (ncio_get_gatt(nf, "MOAD_CEN_LAT", v) == 0) ref_lat = v
    if (ncio_get_gatt(nf, "CEN_LON", v) == 0) ref_lon = v
    if (ncio_get_gatt(nf, "POLE_LAT", v) == 0) pole_lat = v
    if (ncio_get_gatt(nf, "POLE_LON", v) == 0) pole_lon = v
    if (ncio_get_gatt_text(nf, "MAP_PROJ_CHAR", map_proj_char) /= 0) then                     ! model_grid.F90:1288-1295
      select case (proj_code)
      case (PROJ_LC); map_proj_char = "Lambert Conformal"
      case (PROJ_PS); map_proj_char = "Polar Stereographic"
      case (PROJ_MERC); map_proj_char = "Mercator"
      case default; map_proj_char = "Lat/Lon"
      end select
    end if
    call get2("XLONG", "XLONG_M", lon_m, i_target, j_target)
    call get2("XLAT", "XLAT_M", lat_m, i_target, j_target)
    call get2("XLONG_U", "", lon_u, i_target + 1, j_target)
    call get2("XLAT_U", "", lat_u, i_target + 1, j_target)
    call get2("XLONG_V", "", lon_v, i_target, j_target + 1)
    call get2("XLAT_V", "", lat_v, i_target, j_target + 1)
    call get2("MAPFAC_M", "", mapfac_m, i_target, j_target)
    call get2("MAPFAC_U", "", mapfac_u, i_target + 1, j_target)
    call get2("MAPFAC_V", "", mapfac_v, i_target, j_target + 1)
    if (proj_code == PROJ_LC) then
      call get2("SINALPHA", "", sina, i_target, j_target)
      call get2("COSALPHA", "", cosa, i_target, j_target)
    end if
    call ncio_check(ncio_close(nf), "closing "//trim(file_target_grid))
    call get_cell_corners(lat_m, lon_m, lat_c, lon_c)
    call setup_row_block()
    if (nranks > 1) then
      call create_row_block_grid(lat_c, lon_c, grid_h)
    else
      call mpg_check(mpg_grid_create(int(i_target, c_int), int(j_target, c_int), 0_c_int, lon_m, lat_m, lon_c, lat_c, &
                                     lon_u, lat_u, lon_v, lat_v, grid_h), "IN GridCreate")
    end if
    ! The file's projection (MAP_PROJ, TRUELAT1/2, STAND_LON, DX; known point = the grid's own first mass point) as a CLAIM the
    ! library checks on the grid's points: when it holds, the Stores search through the inverse projection instead of the box
    ! pyramid (same weights, a few times faster); a projection that does not reproduce the grid is refused and nothing changes.
    p%code = int(proj_code, c_int)
    p%known_lat = lat_m(1, 1); p%known_lon = lon_m(1, 1); p%known_x = 1.0_dp; p%known_y = 1.0_dp
    p%dx_m = dxkm; p%stand_lon = stand_lon; p%truelat1 = truelat1; p%truelat2 = truelat2
    p%dlat_deg = 0.0_dp; p%dlon_deg = 0.0_dp
    if (proj_code == PROJ_LATLON .and. i_target > 1 .and. j_target > 1) then
      p%dlat_deg = lat_m(1, 2) - lat_m(1, 1); p%dlon_deg = lon_m(2, 1) - lon_m(1, 1)
      if (p%dlon_deg < -180.0_dp) p%dlon_deg = p%dlon_deg + 360.0_dp
    end if
    rc_attach = mpg_grid_attach_proj(grid_h, p, int(je_lo - 1, c_int))
  contains
    subroutine get2(name, alt, a, n1, n2)
      character(len=*), intent(in) :: name, alt
      real(dp), allocatable, intent(out) :: a(:, :)
      integer, intent(in) :: n1, n2
      integer(c_int) :: id
      if (ncio_inq_varid(nf, name, id) /= 0) then
        if (len(alt) == 0) call ncio_check(-1_c_int, "reading "//name//" id")
        call ncio_check(ncio_inq_varid(nf, alt, id), "reading "//name//" / "//alt//" id")
      end if
      allocate (a(n1, n2))
      call ncio_check(ncio_get_var(nf, id, 0_c_int64_t, NCIO_DOUBLE, a), "reading "//name)
    end subroutine get2
  end subroutine define_target_grid_file

  !> get_cell_corners (model_grid.F90:1902-1972) as written: every corner is the great-circle destination point at
  !! distance sqrt(dx**2/2) from a mass point, bearing 135 degrees for the (i, j) block, 225 on the extra column (from
  !! column i_target), 45 on the extra row (from row j_target), 315 at the far corner; its own pi and earth radius.
  subroutine get_cell_corners(lat, lon, latc, lonc)
    real(dp), intent(in) :: lat(:, :), lon(:, :)
    real(dp), allocatable, intent(out) :: latc(:, :), lonc(:, :)
    real(dp), parameter :: pi_gc = 3.14159265359_dp, r_gc = 6370000.0_dp
    real(dp) :: d
    integer :: i, j, ni, nj
    ni = size(lat, 1); nj = size(lat, 2)
    d = sqrt((dxkm**2.0_dp)/2.0_dp)
    allocate (latc(ni + 1, nj + 1), lonc(ni + 1, nj + 1))
    do j = 1, nj
      do i = 1, ni
        call dest(lat(i, j), lon(i, j), 135.0_dp, latc(i, j), lonc(i, j))
      end do
      call dest(lat(ni, j), lon(ni, j), 225.0_dp, latc(ni + 1, j), lonc(ni + 1, j))
    end do
    do i = 1, ni
      call dest(lat(i, nj), lon(i, nj), 45.0_dp, latc(i, nj + 1), lonc(i, nj + 1))
    end do
    call dest(lat(ni, nj), lon(ni, nj), 315.0_dp, latc(ni + 1, nj + 1), lonc(ni + 1, nj + 1))
  contains
    subroutine dest(la, lo, bearing, la2, lo2)
      real(dp), intent(in) :: la, lo, bearing
      real(dp), intent(out) :: la2, lo2
      real(dp) :: lat1, lon1, brng, lat2, lon2
      lat1 = la*(pi_gc/180.0_dp); lon1 = lo*(pi_gc/180.0_dp); brng = bearing*(pi_gc/180.0_dp)
      lat2 = asin(sin(lat1)*cos(d/r_gc) + cos(lat1)*sin(d/r_gc)*cos(brng))
      lon2 = lon1 + atan2(sin(brng)*sin(d/r_gc)*cos(lat1), cos(d/r_gc) - sin(lat1)*sin(lat2))
      la2 = lat2*180.0_dp/pi_gc; lo2 = lon2*180.0_dp/pi_gc
    end subroutine dest
  end subroutine get_cell_corners
end module target_grid

!> Raw-binary named-array container ("MPGRAW1"), the NetCDF stand-in of this build.
!! record = name(32 chars) | dtype int32 (0 = float64, 1 = int32) | ndim int32 | dims 3 x int64 (fastest first) | data
module rawio
  use, intrinsic :: iso_fortran_env, only: int32, int64, real64
  use program_setup, only: fatal
  implicit none
  public
contains
  !> Positions the unit at the data of `name`; returns dims (unused = 1); found = .false. if absent.
  subroutine raw_seek(u, name, dtype, ndim, dims, found)
    integer, intent(in) :: u
    character(len=*), intent(in) :: name
    integer(int32), intent(out) :: dtype, ndim
    integer(int64), intent(out) :: dims(3)
    logical, intent(out) :: found
    character(len=8) :: magic
    character(len=32) :: nm
    integer(int64) :: pos, nbytes
    integer :: ios
    found = .false.
    read (u, pos=1, iostat=ios) magic
    if (ios /= 0 .or. magic /= 'MPGRAW1 ') call fatal("not an MPGRAW1 file", ios)
    pos = 9
    do
      read (u, pos=pos, iostat=ios) nm, dtype, ndim, dims
      if (ios /= 0) return
      pos = pos + 32 + 4 + 4 + 24
      nbytes = dims(1)*dims(2)*dims(3)*merge(8_int64, 4_int64, dtype == 0)
      if (trim(nm) == trim(name)) then
        read (u, pos=pos - 1, iostat=ios) magic(1:1)   ! leave the unit positioned right before the data
        found = .true.
        return
      end if
      pos = pos + nbytes
    end do
  end subroutine raw_seek

  subroutine raw_open_read(file, u)
    character(len=*), intent(in) :: file
    integer, intent(out) :: u
    integer :: ios
    open (newunit=u, file=trim(file), access='stream', form='unformatted', status='old', action='read', iostat=ios)
    if (ios /= 0) call fatal("OPENING INPUT FILE "//trim(file), ios)
  end subroutine raw_open_read

  subroutine raw_open_write(file, u)
    character(len=*), intent(in) :: file
    integer, intent(out) :: u
    integer :: ios
    open (newunit=u, file=trim(file), access='stream', form='unformatted', status='replace', action='write', iostat=ios)
    if (ios /= 0) call fatal("CREATING OUTPUT FILE "//trim(file), ios)
    write (u) 'MPGRAW1 '
  end subroutine raw_open_write

  subroutine raw_write_f64(u, name, ndim, dims, data)
    integer, intent(in) :: u, ndim
    character(len=*), intent(in) :: name
    integer(int64), intent(in) :: dims(3)
    real(real64), intent(in) :: data(*)
    character(len=32) :: nm
    nm = name
    write (u) nm, 0_int32, int(ndim, int32), dims
    write (u) data(1:dims(1)*dims(2)*dims(3))
  end subroutine raw_write_f64
end module rawio
