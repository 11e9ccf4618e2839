!> NetCDF file surface of the Fortran driver, through ncio (ncio_mod.F90: the classic formats, and NetCDF-4 where the library has HDF5):
!!   nc_read_grid     model_grid.F90:287-417   dimensions, lat/lon of cells and vertices, verticesOnCell, ter
!!   nc_load_field    input_data.F90:316-812   one listed variable, first Time record, file order (level fastest)
!!   nc_write_target  write_data.F90:173-1498  dimensions, global attributes, grid variables, every target field with
!!                                             the writer's post-ops (T-300 :1343, MU/PH/P = 0 :1354,:1427,:1466,
!!                                             P_TOP :1362-1371, PB, Z_C :1406-1415, PHB*9.81 :1418), NF90_FLOAT
!! The output is CDF-5 (the reference writes NetCDF-4, an HDF5 container: no libnetcdf in this image, DESIGN.md s7).
module ncfiles
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64
  use ncio
  use mpg
  use program_setup
  use target_grid
  use model_data
  implicit none
  private
  public :: nc_is_netcdf, nc_is_classic, nc_output_format, nc_read_grid, nc_load_field, nc_write_target, nc_read_meta, nc_upload_hgt
  ! what the output header takes from the input files (model_grid.F90:34-46,182; read by input_data.F90:219-245,347-389)
  character(len=50), public :: start_time = ""
  real(dp), public :: config_dt = 0.0_dp
  integer, public :: lsm_scheme = 0, mp_scheme = 0, conv_scheme = 0, diag_out_interval = 0
  character(len=500), public :: nc_in_path = ""      ! path of the file nc_load_field reads from (device flow: raw ranges)

  type fref
    type(field_t), pointer :: p => null()
  end type fref
  type(c_ptr) :: nf_out = c_null_ptr
  character(len=500) :: out_path = ""
  real(dp) :: put_seconds = 0.0_dp             ! wall time inside ncio_put_var (the rest of WRITE DATA is host post-ops)
  integer(c_int) :: d_time, d_we, d_wes, d_sn, d_sns, d_bt, d_bts, d_soil, d_str
  real(dp) :: ptop_vmax = 0.0_dp, ptop_cmin = 0.0_dp       ! this image's reductions for P_TOP (several images)
  integer(c_int) :: ptop_has = 0
  logical :: have_ptop_parts = .false.

contains

  !> global attributes of an open MPAS file that the writer copies into its header.  is_diag: the diag file
  !! (input_data.F90:219-245: start time, config_dt, output_interval); otherwise the history file (:347-389: physics
  !! schemes by name -> WRF option numbers, start time, config_dt).  Missing attributes -> 0, like the reference.
  subroutine nc_read_meta(nf, is_diag)
    type(c_ptr), intent(in) :: nf
    logical, intent(in) :: is_diag
    character(len=64) :: txt
    real(dp) :: v
    if (ncio_get_gatt_text(nf, "config_start_time", txt) == 0) start_time = txt
    config_dt = 0.0_dp
    if (ncio_get_gatt(nf, "config_dt", v) == 0) config_dt = v
    if (is_diag) then
      diag_out_interval = 0
      if (ncio_get_gatt(nf, "output_interval", v) == 0) diag_out_interval = int(v)
      return
    end if
    if (ncio_get_gatt_text(nf, "config_lsm_scheme", txt) /= 0) then
      lsm_scheme = 0
    else if (trim(txt) == 'noah') then
      lsm_scheme = 2
    else if (trim(txt) == 'ruc') then
      lsm_scheme = 3
    end if
    if (ncio_get_gatt_text(nf, "config_microp_scheme", txt) /= 0) then
      mp_scheme = 0
    else if (trim(txt) == 'mp_thompson') then
      mp_scheme = 8
    else if (trim(txt) == 'mp_nssl2m') then
      mp_scheme = 18
    end if
    if (ncio_get_gatt_text(nf, "config_convection_scheme", txt) /= 0) then
      conv_scheme = 0
    else if (trim(txt) == 'cu_ntiedke') then
      conv_scheme = 16
    else if (trim(txt) == 'cu_kain_fritsch') then
      conv_scheme = 1
    else if (trim(txt) == 'cu_grell_freitas') then
      conv_scheme = 3
    end if
  end subroutine nc_read_meta

  !> days since 1970-01-01 of a proleptic Gregorian date (the datetime arithmetic of write_data.F90:1225)
  integer(int64) function days_from_civil(y0, m, d) result(days)
    integer, intent(in) :: y0, m, d
    integer(int64) :: y, era, yoe, doy, doe
    y = y0
    if (m <= 2) y = y - 1
    era = merge(y, y - 399, y >= 0)/400
    yoe = y - era*400
    doy = (153*(m + merge(-3, 9, m > 2)) + 2)/5 + d - 1
    doe = yoe*365 + yoe/4 - yoe/100 + doy
    days = era*146097 + doe - 719468
  end function days_from_civil

  !> seconds of "YYYY-MM-DD_hh:mm:ss" since 1970; ok = .false. when the string does not parse
  integer(int64) function stamp_seconds(s, ok) result(sec)
    character(len=*), intent(in) :: s
    logical, intent(out) :: ok
    integer :: y, mo, d, h, mi, se, ios(6)
    sec = 0
    ok = .false.
    if (len_trim(s) < 19) return
    read (s(1:4), *, iostat=ios(1)) y
    read (s(6:7), *, iostat=ios(2)) mo
    read (s(9:10), *, iostat=ios(3)) d
    read (s(12:13), *, iostat=ios(4)) h
    read (s(15:16), *, iostat=ios(5)) mi
    read (s(18:19), *, iostat=ios(6)) se
    if (any(ios /= 0)) return
    if (mo < 1 .or. mo > 12 .or. d < 1 .or. d > 31) return
    sec = days_from_civil(y, mo, d)*86400_int64 + h*3600 + mi*60 + se
    ok = .true.
  end function stamp_seconds

  !> 'C' = a NetCDF classic file (CDF-1/2/5), 'H' = an HDF5 container (NetCDF-4), ' ' = neither (or unreadable)
  character function nc_magic(file)
    character(len=*), intent(in) :: file
    character(len=4) :: magic
    integer :: u, ios
    nc_magic = ' '
    open (newunit=u, file=trim(file), access='stream', form='unformatted', status='old', iostat=ios)
    if (ios /= 0) return
    read (u, iostat=ios) magic
    close (u)
    if (ios /= 0) return
    if (magic(1:3) == 'CDF') nc_magic = 'C'
    if (magic == achar(137)//'HDF') nc_magic = 'H'
  end function nc_magic

  !> a file ncio reads: the classic formats always, NetCDF-4 where libmpassit_ncio was built with the HDF5 backend (otherwise ncio_open
  !! stops with the way out -- rebuild, or nccopy -k cdf5)
  logical function nc_is_netcdf(file)
    character(len=*), intent(in) :: file
    nc_is_netcdf = nc_magic(file) /= ' '
  end function nc_is_netcdf

  !> a classic file: its variables are byte ranges, which the device-resident flow moves file <-> GPU as they are
  logical function nc_is_classic(file)
    character(len=*), intent(in) :: file
    nc_is_classic = nc_magic(file) == 'C'
  end function nc_is_classic

  !> Format of a ".nc" output: MPASSIT_OUTPUT_FORMAT = cdf5 (default; 64-bit data, device-resident flow), cdf2, or netcdf4 (what the
  !! reference creates, write_data.F90:173 NF90_NETCDF4; through libhdf5 and host arrays).  The namelist has no such switch.
  integer function nc_output_format()
    character(len=32) :: e
    call get_environment_variable("MPASSIT_OUTPUT_FORMAT", e)
    nc_output_format = 5
    select case (trim(e))
    case ("", "cdf5", "CDF5", "5")
      nc_output_format = 5
    case ("cdf2", "CDF2", "2")
      nc_output_format = 2
    case ("netcdf4", "NETCDF4", "nc4", "4")
      nc_output_format = 4
    case default
      call fatal("MPASSIT_OUTPUT_FORMAT must be cdf5, cdf2 or netcdf4 - "//trim(e), 1)
    end select
  end function nc_output_format

  subroutine get_f64(nf, name, arr, n)
    type(c_ptr), intent(in) :: nf
    character(len=*), intent(in) :: name
    integer(int64), intent(in) :: n
    real(dp), allocatable, intent(out) :: arr(:)
    integer(c_int) :: id
    call ncio_check(ncio_inq_varid(nf, name, id), "reading field id - "//trim(name))
    allocate (arr(n))
    call ncio_check(ncio_get_var(nf, id, 0_c_int64_t, NCIO_DOUBLE, arr), "reading field - "//trim(name))
  end subroutine get_f64

  subroutine nc_read_grid(file, latCell, lonCell, latVertex, lonVertex, voc)
    character(len=*), intent(in) :: file
    real(dp), allocatable, intent(out) :: latCell(:), lonCell(:), latVertex(:), lonVertex(:)
    integer(c_int32_t), allocatable, intent(out) :: voc(:)
    type(c_ptr) :: nf
    integer(c_int64_t) :: nc, nv, me, ns
    integer(c_int) :: id
    call ncio_check(ncio_open(file, nf), "opening grid file")
    call ncio_check(ncio_inq_dim(nf, "nCells", nc), "reading nCells")
    call ncio_check(ncio_inq_dim(nf, "nVertices", nv), "reading nVertices")
    call ncio_check(ncio_inq_dim(nf, "maxEdges", me), "reading maxEdges")
    nCells_input = int(nc); nVert_input = int(nv); maxEdges_input = int(me)
    call get_f64(nf, "latCell", latCell, nc)
    call get_f64(nf, "lonCell", lonCell, nc)
    call get_f64(nf, "latVertex", latVertex, nv)
    call get_f64(nf, "lonVertex", lonVertex, nv)
    call ncio_check(ncio_inq_varid(nf, "verticesOnCell", id), "reading verticesOnCell id")
    allocate (voc(nc*me))
    call ncio_check(ncio_get_var(nf, id, 0_c_int64_t, NCIO_INT, voc), "reading verticesOnCell")
    call get_f64(nf, "ter", hgt%src, nc)
    hgt%name = "ter"; hgt%tname = "HGT"; hgt%nlev = 1
    if (ncio_inq_varid(nf, "zs", id) == 0 .and. ncio_inq_dim(nf, "nSoilLevels", ns) == 0) call get_f64(nf, "zs", zs_input, ns)
    call ncio_check(ncio_close(nf), "closing grid file")
  end subroutine nc_read_grid

  !> device flow: `ter` (read whole with the grid) goes up once the source window is known -- its window only
  subroutine nc_upload_hgt()
    integer(c_int64_t) :: i0, n
    if (.not. allocated(hgt%src)) return
    i0 = 0; n = size(hgt%src, kind=c_int64_t)
    if (winn_cell >= 0) then
      i0 = win0_cell; n = winn_cell
    end if
    call mpg_check(mpg_dev_alloc(n*8, hgt%src_dev), "IN dev_alloc ter")
    if (n > 0) call mpg_check(mpg_dev_upload(hgt%src_dev, hgt%src(i0 + 1:i0 + n), n*8), "IN dev_upload ter")
    hgt%src_is_f32 = .false.
    deallocate (hgt%src)
  end subroutine nc_upload_hgt

  subroutine nc_load_field(nf, name, tname, f)
    type(c_ptr), intent(in) :: nf
    character(len=*), intent(in) :: name, tname
    type(field_t), intent(out) :: f
    integer(c_int) :: id, xtype, ndims, isrec, dimids(8)
    integer(c_int64_t) :: shp(8), n
    character(kind=c_char) :: nbuf(64)
    integer :: d0
    f%name = name; f%tname = tname
    call ncio_check(ncio_inq_varid(nf, name, id), "reading field id - "//trim(name))
    call ncio_check(ncio_inq_var(nf, id, nbuf, 64_c_int, xtype, ndims, shp, dimids, isrec), "inquiring "//trim(name))
    d0 = merge(2, 1, isrec /= 0)                  ! skip the Time dimension
    if (ndims - d0 + 1 == 1) then
      f%nlev = 1                                 ! (nCells)
      n = shp(d0)
    else
      f%nlev = int(shp(d0 + 1))                  ! file order [nCells][nlev] == Fortran (nlev, nCells)
      n = shp(d0)*shp(d0 + 1)
    end if
    if (dev_flow) then
      call load_dev(nf, id, xtype, n, shp(d0), f)
      return
    end if
    if (xtype == NCIO_FLOAT) then                 ! stays single precision: the Regrid widens it on the GPU
      allocate (f%src4(n))
      call ncio_check(ncio_get_var(nf, id, 0_c_int64_t, NCIO_FLOAT, f%src4), "reading field - "//trim(name))
    else
      allocate (f%src(n))
      call ncio_check(ncio_get_var(nf, id, 0_c_int64_t, NCIO_DOUBLE, f%src), "reading field - "//trim(name))
    end if
  end subroutine nc_load_field

  !> device flow: the variable's bytes go file -> GPU as stored (big-endian) and the Regrid reads them so (MPG_TYPE_BE);
  !! types other than NF90_FLOAT / NF90_DOUBLE are converted by ncio on the host and uploaded as float64
  !> Only the rows of this image's source window travel: a variable is [nCells | nVertices][levels] in the file, so the
  !! window is ONE byte range of it (the reference has every rank read every variable whole, input_data.F90:645).
  subroutine load_dev(nf, id, xtype, n, nrows, f)
    type(c_ptr), intent(in) :: nf
    integer(c_int), intent(in) :: id, xtype
    integer(c_int64_t), intent(in) :: n, nrows            ! elements of the variable; its leading (cell / vertex) dimension
    type(field_t), intent(inout) :: f
    integer(c_int64_t) :: off, nb, r0, rn, per, es
    real(dp), allocatable :: tmp(:)
    r0 = 0; rn = nrows
    if (nrows == nCells_input .and. winn_cell >= 0) then
      r0 = win0_cell; rn = winn_cell
    else if (nrows == nVert_input .and. nrows /= nCells_input .and. winn_vert >= 0) then
      r0 = win0_vert; rn = winn_vert
    end if
    per = n/max(nrows, 1_c_int64_t)                        ! elements per row (levels)
    if (xtype == NCIO_FLOAT .or. xtype == NCIO_DOUBLE) then
      call ncio_check(ncio_var_extent(nf, id, 0_c_int64_t, off, nb), "locating "//trim(f%name))
      f%src_is_f32 = xtype == NCIO_FLOAT
      es = merge(4, 8, f%src_is_f32)
      call mpg_check(mpg_dev_alloc(rn*per*es, f%src_dev), "IN dev_alloc "//trim(f%name))
      if (rn > 0) call mpg_check(mpg_file_to_dev(trim(nc_in_path), off + r0*per*es, rn*per*es, f%src_dev), "reading field - "//trim(f%name))
      f%src_is_be = .true.
    else
      allocate (tmp(n))
      call ncio_check(ncio_get_var(nf, id, 0_c_int64_t, NCIO_DOUBLE, tmp), "reading field - "//trim(f%name))
      f%src_is_f32 = .false.
      call mpg_check(mpg_dev_alloc(rn*per*8, f%src_dev), "IN dev_alloc "//trim(f%name))
      if (rn > 0) call mpg_check(mpg_dev_upload(f%src_dev, tmp(r0*per + 1:(r0 + rn)*per), rn*per*8), "IN dev_upload "//trim(f%name))
    end if
  end subroutine load_dev

  ! ---- output ---------------------------------------------------------------------------------------------------
  subroutine def_field(name, nlev, stag, id)
    character(len=*), intent(in) :: name
    integer, intent(in) :: nlev, stag            ! stag: 0 mass, 1 U (west_east_stag), 2 V (south_north_stag)
    integer(c_int), intent(out) :: id
    integer(c_int) :: dx, dy, dz
    if (myrank > 0) then                          ! the header is image 0's; the others only need the variable's id
      call ncio_check(ncio_inq_varid(nf_out, name, id), "LOCATING "//trim(name))
      return
    end if
    dx = merge(d_wes, d_we, stag == 1)
    dy = merge(d_sns, d_sn, stag == 2)
    if (nlev == 1) then
      call ncio_check(ncio_def_var(nf_out, name, NCIO_FLOAT, [d_time, dy, dx], id), "DEFINING "//trim(name))
      call ncio_check(ncio_put_att_text(nf_out, id, "MemoryOrder", "XY "), "DEFINING MEMORYORDER")
    else
      if (nlev == nz_input) then
        dz = d_bt
      else if (nlev == nzp1_input) then
        dz = d_bts
      else
        dz = d_soil
      end if
      call ncio_check(ncio_def_var(nf_out, name, NCIO_FLOAT, [d_time, dz, dy, dx], id), "DEFINING "//trim(name))
      call ncio_check(ncio_put_att_text(nf_out, id, "MemoryOrder", "XYZ"), "DEFINING MEMORYORDER")
    end if
    call ncio_check(ncio_put_att_text(nf_out, id, "coordinates", "XLONG XLAT XTIME"), "DEFINING COORD")
    call ncio_check(ncio_put_att_text(nf_out, id, "stagger", merge("X", merge("Y", " ", stag == 2), stag == 1)), "DEFINING STAGGER")
    call ncio_check(ncio_put_att_int(nf_out, id, "FieldType", 104), "DEFINING FieldType")
  end subroutine def_field

  subroutine put_r8(id, a)
    integer(c_int), intent(in) :: id
    real(dp), intent(in) :: a(*)
    integer(int64) :: c0, c1, cr
    call system_clock(c0, cr)
    call ncio_check(ncio_put_var(nf_out, id, 0_c_int64_t, NCIO_DOUBLE, a), "WRITING RECORD")
    call system_clock(c1)
    put_seconds = put_seconds + real(c1 - c0, dp)/real(cr, dp)
  end subroutine put_r8

  !> an all-zero variable (MU, PH, P of wrf_mod_vars: write_data.F90:1354,1427,1466): the record is made to exist
  !! and left as it is in the file just created -- zeros -- instead of converting and writing a gigabyte of them
  subroutine put_zero(id)
    integer(c_int), intent(in) :: id
    integer(c_int64_t) :: off, nb
    if (myrank > 0) return
    if (ncio_format(nf_out) == 4) return      ! NetCDF-4: a record nobody wrote reads as zeros (the fill value) once the file is closed at its record count
    call ncio_check(ncio_var_extent(nf_out, id, 0_c_int64_t, off, nb), "WRITING RECORD")
  end subroutine put_zero

  !> device flow: a float32 field [nlev][rows][nxv] in HBM, big-endian as the Regrid / post-op kernels left it -> the
  !! variable's byte range.  With one image the buffer IS the variable; with
  !! several it holds this image's row block je_lo..je_hi (+1 row on the V stagger) and the owned rows j_lo..j_hi of every
  !! level go to their place in the variable (the last image also owns the top V row).
  subroutine put_dev(id, ptr, nlev, stag)
    integer(c_int), intent(in) :: id
    type(c_ptr), intent(in) :: ptr
    integer, intent(in) :: nlev, stag
    integer(c_int64_t) :: off, nb, n, nxv, ny_buf, ny_glob, jb0, jg0, nrows, k
    integer(int64) :: c0, c1, cr
    call system_clock(c0, cr)
    nxv = i_target + merge(1, 0, stag == 1)
    ny_buf = ny_ext + merge(1, 0, stag == 2)
    n = int(nlev, c_int64_t)*ny_buf*nxv
    call ncio_check(ncio_var_extent(nf_out, id, 0_c_int64_t, off, nb), "WRITING RECORD")
    if (nranks == 1) then
      if (n*4 > nb) call fatal("put_dev: field larger than its variable", int(id))
      call mpg_check(mpg_dev_to_file(trim(out_path), off, n*4, ptr), "WRITING RECORD")
    else
      ny_glob = j_target + merge(1, 0, stag == 2)
      jb0 = j_lo - je_lo
      jg0 = j_lo - 1
      nrows = j_hi - j_lo + 1
      if (stag == 2 .and. myrank == nranks - 1) nrows = nrows + 1
      if (((nlev - 1)*ny_glob + jg0 + nrows)*nxv*4 > nb) call fatal("put_dev: rows beyond the variable", int(id))
      do k = 0, nlev - 1
        call mpg_check(mpg_dev_to_file(trim(out_path), off + ((k*ny_glob + jg0)*nxv)*4, nrows*nxv*4, &
                                       ptr_add(ptr, ((k*ny_buf + jb0)*nxv)*4)), "WRITING RECORD")
      end do
    end if
    call system_clock(c1)
    put_seconds = put_seconds + real(c1 - c0, dp)/real(cr, dp)
  end subroutine put_dev

  type(c_ptr) function ptr_add(p, nbytes)
    type(c_ptr), intent(in) :: p
    integer(c_int64_t), intent(in) :: nbytes
    ptr_add = transfer(transfer(p, 0_c_intptr_t) + int(nbytes, c_intptr_t), p)
  end function ptr_add

  !> one target field of the device flow with the writer's post-ops as device epilogues (write_data.F90:1339-1475)
  subroutine write_field_dev(p, id, id_extra, id_ptop, npts)
    type(field_t), intent(inout) :: p
    integer(c_int), intent(in) :: id, id_extra(8), id_ptop
    integer, intent(in) :: npts                   ! points of this image's row block on the mass stagger
    type(c_ptr) :: tmp
    real(dp) :: ptop
    integer(c_int64_t) :: n
    n = p%n_dst_elems
    if (p%dst_is_f32) then
      if (.not. p%dst_is_be) call fatal("write_field_dev: float32 field in host byte order - "//trim(p%tname), -1)
      call put_dev(id, p%dst_dev, p%nlev, p%stagger)
    else
      call mpg_check(mpg_dev_alloc(n*4, tmp), "IN dev_alloc")
      if (trim(p%tname) == 'PHB') then
        call mpg_check(mpg_post_layer_mean_dev(p%dst_dev, int(p%nlev, c_int), int(npts, c_int64_t), tmp, 1_c_int, c_null_ptr), "IN Z_C")   ! :1406-1415
        call put_dev(id_extra(3), tmp, p%nlev - 1, 0)
        call mpg_check(mpg_post_cast_dev(p%dst_dev, n, 9.81_c_double, 0.0_c_double, tmp, 1_c_int, c_null_ptr), "IN PHB*9.81")     ! :1418
        call put_dev(id, tmp, p%nlev, p%stagger)
        if (wrf_mod_vars) call put_zero(id_extra(4))
      else
        if (wrf_mod_vars .and. trim(p%tname) == 'P_HYD') then                                                                    ! :1362-1379
          if (nranks == 1) then
            call mpg_check(mpg_post_ptop_dev(p%dst_dev, int(p%nlev, c_int), int(npts, c_int64_t), ptop, c_null_ptr), "IN P_TOP")
            call ncio_check(ncio_put_var(nf_out, id_ptop, 0_c_int64_t, NCIO_DOUBLE, [ptop]), "WRITING P_TOP")
          else   ! the block's two reductions; image 0 combines them once every image has reported (finish_ranks)
            call mpg_check(mpg_post_ptop_parts_dev(p%dst_dev, int(p%nlev, c_int), int(npts, c_int64_t), ptop_vmax, ptop_cmin, ptop_has, &
                                                   c_null_ptr), "IN P_TOP")
            have_ptop_parts = .true.
          end if
        end if
        call mpg_check(mpg_post_cast_dev(p%dst_dev, n, 1.0_c_double, 0.0_c_double, tmp, 1_c_int, c_null_ptr), "IN cast")
        call put_dev(id, tmp, p%nlev, p%stagger)
        if (wrf_mod_vars .and. trim(p%tname) == 'P_HYD') call put_dev(id_extra(2), tmp, p%nlev, p%stagger)                       ! PB = P_HYD
      end if
      call mpg_check(mpg_dev_free(tmp), "IN dev_free")
    end if
    if (wrf_mod_vars .and. trim(p%tname) == 'MUB') call put_zero(id_extra(1))
    call mpg_check(mpg_dev_free(p%dst_dev), "IN dev_free")
    p%dst_dev = c_null_ptr
  end subroutine write_field_dev

  ! ---- several driver images, one output file -----------------------------------------------------------------------
  ! Image 0 creates the file, writes the header, the grid and time variables and its own rows, closes it and raises
  ! <output>.<run id>.ready; the other images (which have regridded their blocks meanwhile) then write their rows straight
  ! into the variables' byte ranges and report <output>.<run id>.done.<rank> with their two P_TOP reductions; image 0 waits
  ! for all of them, stores P_TOP and removes the marker files.  No data moves between the images.
  function marker_name(kind, rank) result(name)
    character(len=*), intent(in) :: kind
    integer, intent(in) :: rank
    character(len=600) :: name
    if (rank >= 0) then
      write (name, '(a,".",a,".",a,".",i0)') trim(out_path), trim(run_id), kind, rank
    else
      write (name, '(a,".",a,".",a)') trim(out_path), trim(run_id), kind
    end if
  end function marker_name

  subroutine wait_for(file)
    character(len=*), intent(in) :: file
    logical :: there
    integer(int64) :: c0, c1, cr
    real(dp) :: limit
    character(len=32) :: e
    character(len=200) :: msg
    integer :: ios
    limit = 1800.0_dp                              ! MPASSIT_WAIT_S: how long an image waits for another's marker
    call get_environment_variable("MPASSIT_WAIT_S", e)
    if (len_trim(e) > 0) then
      read (e, *, iostat=ios) limit
      if (ios /= 0 .or. limit <= 0.0_dp) limit = 1800.0_dp
    end if
    call system_clock(c0, cr)
    do
      inquire (file=trim(file), exist=there)
      if (there) exit
      call system_clock(c1)
      if (real(c1 - c0, dp)/real(cr, dp) > limit) then
        write (msg, '(a,i0,a,i0,a,i0,a)') "image ", myrank, " of ", nranks, " waited ", int(limit), &
          " s for another image's marker (are all images running? MPASSIT_NRANKS, or mpiexec's / srun's rank variables, say how many there are) - "
        call fatal(trim(msg)//trim(file), -1)
      end if
      if (ncio_msleep(2_c_int) /= 0) exit
    end do
  end subroutine wait_for

  !> the ready marker holds the image count and the size of the file image 0 closed: a marker that does not describe THIS
  !! run's file (left by another run, or a file truncated since) stops the image instead of letting it write into it
  subroutine check_ready_marker()
    real(dp) :: vals(2)
    integer :: u, ios
    integer(int64) :: fsize
    open (newunit=u, file=trim(marker_name("ready", -1)), form='formatted', status='old', action='read', iostat=ios)
    if (ios /= 0) call fatal("cannot read "//trim(marker_name("ready", -1)), ios)
    read (u, *, iostat=ios) vals
    close (u)
    inquire (file=trim(out_path), size=fsize)
    if (ios /= 0 .or. nint(vals(1)) /= nranks .or. int(vals(2), int64) /= fsize) &
      call fatal("the ready marker does not match this run's output file (stale marker?) - "//trim(out_path), myrank)
  end subroutine check_ready_marker

  subroutine write_marker(file, vals)
    character(len=*), intent(in) :: file
    real(dp), intent(in) :: vals(:)
    integer :: u
    ! written under a temporary name and renamed: a reader never sees a half-written marker
    open (newunit=u, file=trim(file)//".tmp", form='formatted', status='replace', action='write')
    write (u, '(4es26.17e3)') vals
    close (u)
    call rename_file(trim(file)//".tmp", trim(file))
  end subroutine write_marker

  subroutine rename_file(a, b)
    character(len=*), intent(in) :: a, b
    if (ncio_rename(a, b) /= 0) call fatal("renaming "//a, -1)
  end subroutine rename_file

  subroutine remove_file(file)
    character(len=*), intent(in) :: file
    integer :: u, ios
    open (newunit=u, file=trim(file), status='old', iostat=ios)
    if (ios == 0) close (u, status='delete')
  end subroutine remove_file

  !> after this image's rows are in the file
  subroutine finish_ranks(id_ptop_var, have_ptop)
    integer(c_int), intent(in) :: id_ptop_var
    logical, intent(in) :: have_ptop
    real(dp) :: vals(3), vmax, cmin
    logical :: any_c
    integer :: r, u
    integer(c_int64_t) :: off, nb
    integer(c_int8_t) :: b4(4)
    real(c_float) :: ptop4
    type(c_ptr) :: nf_r
    if (myrank > 0) then
      call write_marker(marker_name("done", myrank), [ptop_vmax, ptop_cmin, real(ptop_has, dp)])
      return
    end if
    vmax = ptop_vmax; cmin = ptop_cmin; any_c = ptop_has /= 0
    do r = 1, nranks - 1
      call wait_for(marker_name("done", r))
      open (newunit=u, file=trim(marker_name("done", r)), form='formatted', status='old', action='read')
      read (u, *) vals
      close (u)
      vmax = max(vmax, vals(1))
      if (vals(3) /= 0.0_dp) then
        if (any_c) then
          cmin = min(cmin, vals(2))
        else
          cmin = vals(2)
        end if
        any_c = .true.
      end if
    end do
    if (have_ptop .and. have_ptop_parts) then      ! write_data.F90:1362-1371 on the combined reductions
      ptop4 = real(vmax, c_float)
      if (any_c) ptop4 = real(min(vmax, cmin), c_float)
      call ncio_check(ncio_open(trim(out_path), nf_r), "reopening "//trim(out_path))
      call ncio_check(ncio_var_extent(nf_r, id_ptop_var, 0_c_int64_t, off, nb), "locating P_TOP")
      call ncio_check(ncio_close(nf_r), "closing "//trim(out_path))
      b4 = transfer(ptop4, b4)
      b4 = b4(4:1:-1)                              ! NetCDF classic stores big-endian
      open (newunit=u, file=trim(out_path), access='stream', form='unformatted', status='old', action='readwrite')
      write (u, pos=off + 1) b4
      close (u)
    end if
    do r = 1, nranks - 1
      call remove_file(marker_name("done", r))
    end do
    call remove_file(marker_name("ready", -1))
  end subroutine finish_ranks

  subroutine put_r4(id, a)
    integer(c_int), intent(in) :: id
    real(c_float), intent(in) :: a(*)
    integer(int64) :: c0, c1, cr
    call system_clock(c0, cr)
    call ncio_check(ncio_put_var(nf_out, id, 0_c_int64_t, NCIO_FLOAT, a), "WRITING RECORD")
    call system_clock(c1)
    put_seconds = put_seconds + real(c1 - c0, dp)/real(cr, dp)
  end subroutine put_r4

  subroutine gatt_t(name, text)
    character(len=*), intent(in) :: name, text
    call ncio_check(ncio_put_att_text(nf_out, NCIO_GLOBAL, name, text), "DEFINING "//name//" GLOBAL ATTRIBUTE")
  end subroutine gatt_t
  subroutine gatt_i(name, val)
    character(len=*), intent(in) :: name
    integer, intent(in) :: val
    call ncio_check(ncio_put_att_int(nf_out, NCIO_GLOBAL, name, val), "DEFINING "//name//" GLOBAL ATTRIBUTE")
  end subroutine gatt_i
  subroutine gatt_r(name, val)
    character(len=*), intent(in) :: name
    real(dp), intent(in) :: val
    call ncio_check(ncio_put_att_real(nf_out, NCIO_GLOBAL, name, val), "DEFINING "//name//" GLOBAL ATTRIBUTE")
  end subroutine gatt_r

  subroutine nc_write_target(file, valid_time)
    character(len=*), intent(in) :: file, valid_time
    integer, parameter :: MAXV = 512
    integer(c_int) :: ids(MAXV), id_extra(8), id_grid(8), id_mf(3), id_zs, id_times, id_ptop, id_itime, id_xtime
    character(len=19) :: st
    integer(int64) :: xt_sec
    integer(c_int32_t) :: itime(1)
    logical :: s_ok, v_ok
    real(dp), allocatable :: zs(:)
    integer :: nv, i, k, npts
    type(fref), allocatable :: fl(:)
    real(dp), allocatable :: tmp(:)
    real(dp) :: ptop
    character(len=19) :: tstr
    integer(c_int8_t) :: tbytes(19)
    logical :: have_ptop
    integer(int64) :: closed_size
    ! nz / nzp1 / nsoil from what was read
    if (do_u_interp == 1) nz_input = u_field%nlev
    if (nz_input == 0 .and. hist_3d_nz%n > 0) nz_input = hist_3d_nz%f(1)%nlev
    if (nz_input == 0 .and. diag_bundle%n > 0) nz_input = maxval(diag_bundle%f(1:diag_bundle%n)%nlev)
    nz_input = max(nz_input, 1)
    nzp1_input = nz_input + 1
    if (hist_soil%n > 0) nsoil_input = hist_soil%f(1)%nlev
    nsoil_input = max(nsoil_input, 1)
    out_path = file
    if (nranks > 1 .and. .not. dev_flow) call fatal("several driver images need NetCDF CLASSIC files in and out (the device-resident flow; NetCDF-4 goes through one image)", nranks)
    if (myrank > 0) then
      ! image 0 has created the file, written header, grid and time variables and its own rows, and closed it
      call wait_for(marker_name("ready", -1))
      call check_ready_marker()
      call ncio_check(ncio_open(trim(file), nf_out), "opening "//trim(file))
    else
      call remove_file(marker_name("ready", -1))
    call ncio_check(ncio_create(file, nc_output_format(), nf_out), "CREATING FILE "//trim(file))
    call ncio_check(ncio_def_dim(nf_out, "Time", 0, d_time), "DEFINING Time")                    ! write_data.F90:177-194
    call ncio_check(ncio_def_dim(nf_out, "west_east", i_target, d_we), "DEFINING west_east")
    call ncio_check(ncio_def_dim(nf_out, "west_east_stag", i_target + 1, d_wes), "DEFINING west_east_stag")
    call ncio_check(ncio_def_dim(nf_out, "south_north", j_target, d_sn), "DEFINING south_north")
    call ncio_check(ncio_def_dim(nf_out, "south_north_stag", j_target + 1, d_sns), "DEFINING south_north_stag")
    call ncio_check(ncio_def_dim(nf_out, "bottom_top", nz_input, d_bt), "DEFINING bottom_top")
    call ncio_check(ncio_def_dim(nf_out, "bottom_top_stag", nzp1_input, d_bts), "DEFINING bottom_top_stag")
    call ncio_check(ncio_def_dim(nf_out, "soil_layers_stag", nsoil_input, d_soil), "DEFINING soil_layers_stag")
    call ncio_check(ncio_def_dim(nf_out, "StrLen", 19, d_str), "DEFINING StrLen")
    call ncio_check(ncio_put_att_int(nf_out, NCIO_GLOBAL, "WEST-EAST_GRID_DIMENSION", i_target + 1), "GLOBAL ATT")   ! :196-308
    call ncio_check(ncio_put_att_int(nf_out, NCIO_GLOBAL, "SOUTH-NORTH_GRID_DIMENSION", j_target + 1), "GLOBAL ATT")
    call ncio_check(ncio_put_att_int(nf_out, NCIO_GLOBAL, "BOTTOM-TOP_GRID_DIMENSION", nz_input + 1), "GLOBAL ATT")
    st = start_time(1:19)
    if (len_trim(start_time) == 0) st = valid_time
    call gatt_t("SIMULATION_START_DATE", st)
    call gatt_t("START_DATE", st)
    call gatt_r("DX", dxkm)
    call gatt_r("DY", dxkm)
    call gatt_r("DT", config_dt)
    call gatt_i("SF_SURFACE_PHYSICS", lsm_scheme)
    call gatt_i("MP_PHYSICS", mp_scheme)
    call gatt_i("CU_PHYSICS", conv_scheme)
    call gatt_r("CEN_LAT", ref_lat)
    call gatt_r("CEN_LON", ref_lon)
    call gatt_r("TRUELAT1", truelat1)
    call gatt_r("TRUELAT2", truelat2)
    call gatt_r("MOAD_CEN_LAT", ref_lat)
    call gatt_r("STAND_LON", stand_lon)
    call gatt_r("POLE_LAT", pole_lat)
    call gatt_r("POLE_LON", pole_lon)
    call gatt_r("POL_ELAT", pole_lat)                                                            ! sic (:253)
    call gatt_i("MAP_PROJ", proj_code)
    call gatt_t("MAP_PROJ_CHAR", trim(map_proj_char))
    if (interp_diag) call gatt_i("PREC_ACC_DT", diag_out_interval)                               ! :262-265
    call gatt_i("I_PARENT_START", 1)
    call gatt_i("J_PARENT_START", 1)
    call gatt_i("WEST-EAST_PATCH_START_UNSTAG", 1)
    call gatt_i("WEST-EAST_PATCH_START_STAG", 1)
    call gatt_i("SOUTH-NORTH_PATCH_START_UNSTAG", 1)
    call gatt_i("SOUTH-NORTH_PATCH_START_STAG", 1)
    call gatt_i("BOTTOM-TOP_PATCH_START_UNSTAG", 1)
    call gatt_i("BOTTOM-TOP_PATCH_START_STAG", 1)
    call gatt_i("WEST-EAST_PATCH_END_UNSTAG", i_target)
    call gatt_i("WEST-EAST_PATCH_END_STAG", i_target + 1)
    call gatt_i("SOUTH-NORTH_PATCH_END_UNSTAG", j_target)
    call gatt_i("SOUTH-NORTH_PATCH_END_STAG", j_target + 1)
    call gatt_i("BOTTOM-TOP_PATCH_END_UNSTAG", nz_input)
    call gatt_i("BOTTOM-TOP_PATCH_END_STAG", nz_input + 1)
    end if
    ! grid variables (:312-476): XLONG, XLAT on the three staggers, SINALPHA / COSALPHA for Lambert
    call def_field("XLONG", 1, 0, id_grid(1)); call def_field("XLAT", 1, 0, id_grid(2))
    call def_field("XLONG_U", 1, 1, id_grid(3)); call def_field("XLAT_U", 1, 1, id_grid(4))
    call def_field("XLONG_V", 1, 2, id_grid(5)); call def_field("XLAT_V", 1, 2, id_grid(6))
    if (proj_code == PROJ_LC) then
      call def_field("SINALPHA", 1, 0, id_grid(7)); call def_field("COSALPHA", 1, 0, id_grid(8))
    end if
    call def_field("MAPFAC_M", 1, 0, id_mf(1)); call def_field("MAPFAC_U", 1, 1, id_mf(2)); call def_field("MAPFAC_V", 1, 2, id_mf(3))
    if (myrank == 0) then
    call ncio_check(ncio_def_var(nf_out, "ZS", NCIO_FLOAT, [d_time, d_soil], id_zs), "DEFINING ZS")
    call ncio_check(ncio_def_var(nf_out, "Times", NCIO_CHAR, [d_time, d_str], id_times), "DEFINING Times")          ! :522-534
    call ncio_check(ncio_put_att_text(nf_out, id_times, "description", "Times"), "DEFINING Times NAME")
    call ncio_check(ncio_put_att_text(nf_out, id_times, "units", "m"), "DEFINING Times UNITS")
    call ncio_check(ncio_put_att_text(nf_out, id_times, "coordinates", "Time"), "DEFINING Times COORD")
    call ncio_check(ncio_put_att_text(nf_out, id_times, "stagger", ""), "DEFINING STAGGER")
    call ncio_check(ncio_put_att_int(nf_out, id_times, "FieldType", 104), "DEFINING FieldType")
    call ncio_check(ncio_def_var(nf_out, "ITIMESTEP", NCIO_INT, [d_time], id_itime), "DEFINING ITIMESTEP")              ! :537-548
    call ncio_check(ncio_put_att_text(nf_out, id_itime, "description", ""), "DEFINING ITIMESTEP NAME")
    call ncio_check(ncio_put_att_text(nf_out, id_itime, "units", ""), "DEFINING ITIMESTEP UNITS")
    call ncio_check(ncio_put_att_text(nf_out, id_itime, "stagger", ""), "DEFINING STAGGER")
    call ncio_check(ncio_put_att_int(nf_out, id_itime, "FieldType", 106), "DEFINING FieldType")
    call ncio_check(ncio_put_att_text(nf_out, id_itime, "MemoryOrder", "O "), "DEFINING MemoryOrder")
    call ncio_check(ncio_def_var(nf_out, "XTIME", NCIO_FLOAT, [d_time], id_xtime), "DEFINING XTIME")                   ! :550-561
    call ncio_check(ncio_put_att_text(nf_out, id_xtime, "description", "minutes since "//st), "DEFINING XTIME NAME")
    call ncio_check(ncio_put_att_text(nf_out, id_xtime, "units", "minutes since "//st), "DEFINING XTIME UNITS")
    call ncio_check(ncio_put_att_text(nf_out, id_xtime, "stagger", ""), "DEFINING STAGGER")
    call ncio_check(ncio_put_att_int(nf_out, id_xtime, "FieldType", 104), "DEFINING FieldType")
    call ncio_check(ncio_put_att_text(nf_out, id_xtime, "MemoryOrder", "O "), "DEFINING MemoryOrder")
    end if
    ! target fields in the writer's order (:1150-1475)
    call collect(fl, nv)
    if (nv > MAXV) call fatal("too many output variables", nv)
    have_ptop = .false.
    id_extra = -1
    do i = 1, nv
      call def_field(trim(fl(i)%p%tname), fl(i)%p%nlev, fl(i)%p%stagger, ids(i))
      if (wrf_mod_vars .and. trim(fl(i)%p%tname) == 'MUB') call def_field("MU", fl(i)%p%nlev, 0, id_extra(1))
      if (wrf_mod_vars .and. trim(fl(i)%p%tname) == 'P_HYD') then
        if (myrank == 0) then
          call ncio_check(ncio_def_var(nf_out, "P_TOP", NCIO_FLOAT, [d_time], id_ptop), "DEFINING P_TOP")
        else
          call ncio_check(ncio_inq_varid(nf_out, "P_TOP", id_ptop), "LOCATING P_TOP")
        end if
        call def_field("PB", fl(i)%p%nlev, 0, id_extra(2))
        have_ptop = .true.
      end if
      if (trim(fl(i)%p%tname) == 'PHB') then
        call def_field("Z_C", nzp1_input, 0, id_extra(3))                                        ! on bottom_top_stag (:479)
        if (wrf_mod_vars) call def_field("PH", fl(i)%p%nlev, 0, id_extra(4))
      end if
    end do
    if (wrf_mod_vars .and. hist_3d_nz%n > 0) call def_field("P", nz_input, 0, id_extra(5))
    if (myrank == 0) then
    call ncio_check(ncio_enddef(nf_out), "ENDDEF")
    ! ---- data ----
    call put_r8(id_grid(1), lon_m); call put_r8(id_grid(2), lat_m)
    call put_r8(id_grid(3), lon_u); call put_r8(id_grid(4), lat_u)
    call put_r8(id_grid(5), lon_v); call put_r8(id_grid(6), lat_v)
    if (proj_code == PROJ_LC) then
      call put_r8(id_grid(7), sina); call put_r8(id_grid(8), cosa)
    end if
    call put_r8(id_mf(1), mapfac_m); call put_r8(id_mf(2), mapfac_u); call put_r8(id_mf(3), mapfac_v)
    allocate (zs(nsoil_input)); zs = 0.0_dp
    if (allocated(zs_input)) zs(1:min(size(zs_input), nsoil_input)) = zs_input(1:min(size(zs_input), nsoil_input))
    call put_r8(id_zs, zs)
    tstr = valid_time
    tbytes = transfer(tstr, tbytes)
    call ncio_check(ncio_put_var(nf_out, id_times, 0_c_int64_t, NCIO_CHAR, tbytes), "WRITING Times")
    ! XTIME = datetime(start) - datetime(valid) in minutes, in THIS order as the reference has it (:1225-1227: a valid time
    ! after the start gives a negative value); ITIMESTEP = int(seconds / config_dt), 0 without a time step (:1233-1240)
    xt_sec = 0
    s_ok = .false.; v_ok = .false.
    xt_sec = stamp_seconds(st, s_ok) - stamp_seconds(tstr, v_ok)
    if (.not. (s_ok .and. v_ok)) xt_sec = 0
    call ncio_check(ncio_put_var(nf_out, id_xtime, 0_c_int64_t, NCIO_DOUBLE, [real(xt_sec, dp)/60.0_dp]), "WRITING XTIME RECORD")
    itime(1) = 0
    if (config_dt > 0.0_dp) itime(1) = int(real(xt_sec, dp)/config_dt, c_int32_t)
    call ncio_check(ncio_put_var(nf_out, id_itime, 0_c_int64_t, NCIO_INT, itime), "WRITING ITIMESTEP RECORD")
    if (have_ptop .and. nranks > 1) call ncio_check(ncio_put_var(nf_out, id_ptop, 0_c_int64_t, NCIO_DOUBLE, [0.0_dp]), "WRITING P_TOP")
    end if
    npts = i_target*ny_ext                              ! this image's row block (the whole grid with one image)
    do i = 1, nv
      if (dev_flow) then
        call write_field_dev(fl(i)%p, ids(i), id_extra, id_ptop, npts)
        cycle
      end if
      if (allocated(fl(i)%p%dst4)) then                                                          ! came back as NF90_FLOAT, post-op fused
        call put_r4(ids(i), fl(i)%p%dst4)
      else if (wrf_mod_vars .and. trim(fl(i)%p%tname) == 'T') then
        tmp = fl(i)%p%dst - 300.0_dp                                                               ! :1339-1347
        call put_r8(ids(i), tmp)
      else if (trim(fl(i)%p%tname) == 'PHB') then
        if (allocated(tmp)) deallocate (tmp)
        allocate (tmp(npts*nzp1_input))
        tmp = 0.0_dp
        do k = 2, fl(i)%p%nlev                                                                     ! :1406-1412
          tmp((k - 2)*npts + 1:(k - 1)*npts) = 0.5_dp*(fl(i)%p%dst((k - 1)*npts + 1:k*npts) + fl(i)%p%dst((k - 2)*npts + 1:(k - 1)*npts))
        end do
        call put_r8(id_extra(3), tmp)
        tmp = fl(i)%p%dst*9.81_dp                                                                  ! :1418
        call put_r8(ids(i), tmp(1:size(fl(i)%p%dst)))
        if (wrf_mod_vars) call put_zero(id_extra(4))
        deallocate (tmp)
      else
        call put_r8(ids(i), fl(i)%p%dst)
      end if
      if (wrf_mod_vars .and. trim(fl(i)%p%tname) == 'MUB') call put_zero(id_extra(1))
      if (wrf_mod_vars .and. trim(fl(i)%p%tname) == 'P_HYD') then                                  ! :1362-1379
        ptop = maxval(fl(i)%p%dst)
        do k = (fl(i)%p%nlev - 1)*npts + 1, fl(i)%p%nlev*npts
          if (fl(i)%p%dst(k) >= 10.0_dp) ptop = min(fl(i)%p%dst(k)*0.80_dp, ptop)
        end do
        call ncio_check(ncio_put_var(nf_out, id_ptop, 0_c_int64_t, NCIO_DOUBLE, [ptop]), "WRITING P_TOP")
        call put_r8(id_extra(2), fl(i)%p%dst)
      end if
    end do
    if (id_extra(5) >= 0) call put_zero(id_extra(5))
    call ncio_check(ncio_close(nf_out), "CLOSING FILE")
    if (nranks > 1) then
      if (myrank == 0) then
        inquire (file=trim(out_path), size=closed_size)
        call write_marker(marker_name("ready", -1), [real(nranks, dp), real(closed_size, dp)])
      end if
      call finish_ranks(id_ptop, have_ptop)
    end if
    print '(a,f9.3,a)', "   [WRITE DATA: of which inside ncio_put_var] ", put_seconds, " s"
  end subroutine nc_write_target

  !> every target field in the order write_target_data emits them (references, no copies)
  subroutine collect(fl, nv)
    type(fref), allocatable, intent(out) :: fl(:)
    integer, intent(out) :: nv
    integer :: cap
    cap = 8 + diag_bundle%n + hist_2d_patch%n + hist_2d_cons%n + hist_2d_nstd%n + hist_3d_nz%n + hist_3d_nzp1%n + hist_3d_vert%n + hist_soil%n
    allocate (fl(cap))
    nv = 0
    if (interp_hist) then
      call add(hgt, 0)
      if (do_u_interp == 1) call add(u_field, 1)
      if (do_v_interp == 1) call add(v_field, 2)
    end if
    if (interp_diag) call add_bundle(diag_bundle, 1)          ! 2-D diag fields first (:584)
    if (interp_hist) then
      call add_bundle(hist_2d_cons, 0); call add_bundle(hist_2d_patch, 0); call add_bundle(hist_2d_nstd, 0)
    end if
    if (interp_diag) call add_bundle(diag_bundle, 2)          ! then the 3-D diag fields (:604)
    if (interp_hist) then
      call add_bundle(hist_soil, 0); call add_bundle(hist_3d_nz, 0); call add_bundle(hist_3d_nzp1, 0); call add_bundle(hist_3d_vert, 0)
    end if
  contains
    subroutine add(f, stag)
      type(field_t), intent(inout), target :: f
      integer, intent(in) :: stag
      if (.not. allocated(f%dst) .and. .not. allocated(f%dst4) .and. .not. c_associated(f%dst_dev)) return
      nv = nv + 1
      f%stagger = stag
      fl(nv)%p => f
    end subroutine add
    subroutine add_bundle(b, sel)
      type(bundle_t), intent(inout), target :: b
      integer, intent(in) :: sel                 ! 0 all, 1 only 2-D, 2 only 3-D
      integer :: q
      do q = 1, b%n
        if (sel == 1 .and. b%f(q)%nlev /= 1) cycle
        if (sel == 2 .and. b%f(q)%nlev == 1) cycle
        call add(b%f(q), 0)
      end do
    end subroutine add_bundle
  end subroutine collect
end module ncfiles
