!> ISO_C_BINDING interface of the host-side NetCDF classic I/O (include/mpassit_ncio.h, hostio/ncclassic.c):
!! the nf90_* calls of model_grid.F90:287-417, input_data.F90:145-812 and write_data.F90:173-1498 map one to one
!! onto these (nf90_open -> ncio_open, nf90_inq_varid -> ncio_inq_varid, nf90_get_var -> ncio_get_var, ...).
module ncio
  use, intrinsic :: iso_c_binding
  implicit none
  public
  integer(c_int), parameter :: NCIO_CHAR = 2, NCIO_INT = 4, NCIO_FLOAT = 5, NCIO_DOUBLE = 6, NCIO_GLOBAL = -1

  interface
    function ncio_strerror_c() bind(C, name="ncio_strerror") result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function ncio_strerror_c
    function ncio_open_c(path, f) bind(C, name="ncio_open") result(rc)
      import :: c_char, c_ptr, c_int
      character(kind=c_char), intent(in) :: path(*)
      type(c_ptr), intent(out) :: f
      integer(c_int) :: rc
    end function ncio_open_c
    function ncio_inq_dim_c(f, name, len, is_unl) bind(C, name="ncio_inq_dim") result(rc)
      import :: c_char, c_ptr, c_int, c_int64_t
      type(c_ptr), value :: f
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int64_t), intent(out) :: len
      integer(c_int), intent(out) :: is_unl
      integer(c_int) :: rc
    end function ncio_inq_dim_c
    function ncio_inq_varid_c(f, name, varid) bind(C, name="ncio_inq_varid") result(rc)
      import :: c_char, c_ptr, c_int
      type(c_ptr), value :: f
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: varid
      integer(c_int) :: rc
    end function ncio_inq_varid_c
    function ncio_inq_var(f, varid, name_buf, buf_len, xtype, ndims, shape, dimids, is_record) bind(C, name="ncio_inq_var") result(rc)
      import :: c_char, c_ptr, c_int, c_int64_t
      type(c_ptr), value :: f
      integer(c_int), value :: varid, buf_len
      character(kind=c_char) :: name_buf(*)
      integer(c_int), intent(out) :: xtype, ndims, is_record
      integer(c_int64_t), intent(out) :: shape(*)
      integer(c_int), intent(out) :: dimids(*)
      integer(c_int) :: rc
    end function ncio_inq_var
    function ncio_get_var(f, varid, rec, mem_type, buf) bind(C, name="ncio_get_var") result(rc)
      import :: c_ptr, c_int, c_int64_t
      type(c_ptr), value :: f
      integer(c_int), value :: varid, mem_type
      integer(c_int64_t), value :: rec
      type(*), dimension(*) :: buf
      integer(c_int) :: rc
    end function ncio_get_var
    function ncio_get_att_double_c(f, varid, name, vals, max_n, n) bind(C, name="ncio_get_att_double") result(rc)
      import :: c_char, c_ptr, c_int, c_double
      type(c_ptr), value :: f
      integer(c_int), value :: varid, max_n
      character(kind=c_char), intent(in) :: name(*)
      real(c_double), intent(out) :: vals(*)
      integer(c_int), intent(out) :: n
      integer(c_int) :: rc
    end function ncio_get_att_double_c
    !> byte range of a variable / record inside the file; on a file being written the range is made to exist
    !! (a fresh file reads as zeros there)
    function ncio_var_extent(f, varid, rec, offset, nbytes) bind(C, name="ncio_var_extent") result(rc)
      import :: c_ptr, c_int, c_int64_t
      type(c_ptr), value :: f
      integer(c_int), value :: varid
      integer(c_int64_t), value :: rec
      integer(c_int64_t), intent(out) :: offset, nbytes
      integer(c_int) :: rc
    end function ncio_var_extent
    !> 1, 2, 5: a classic file; 4: NetCDF-4 (an HDF5 container: no byte ranges, ncio_var_extent refuses)
    function ncio_format(f) bind(C, name="ncio_format") result(fmt)
      import :: c_ptr, c_int
      type(c_ptr), value :: f
      integer(c_int) :: fmt
    end function ncio_format
    function ncio_create_c(path, fmt, f) bind(C, name="ncio_create") result(rc)
      import :: c_char, c_ptr, c_int
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), value :: fmt
      type(c_ptr), intent(out) :: f
      integer(c_int) :: rc
    end function ncio_create_c
    function ncio_def_dim_c(f, name, len, dimid) bind(C, name="ncio_def_dim") result(rc)
      import :: c_char, c_ptr, c_int, c_int64_t
      type(c_ptr), value :: f
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int64_t), value :: len
      integer(c_int), intent(out) :: dimid
      integer(c_int) :: rc
    end function ncio_def_dim_c
    function ncio_def_var_c(f, name, xtype, ndims, dimids, varid) bind(C, name="ncio_def_var") result(rc)
      import :: c_char, c_ptr, c_int
      type(c_ptr), value :: f
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), value :: xtype, ndims
      integer(c_int), intent(in) :: dimids(*)
      integer(c_int), intent(out) :: varid
      integer(c_int) :: rc
    end function ncio_def_var_c
    function ncio_put_att_text_c(f, varid, name, text) bind(C, name="ncio_put_att_text") result(rc)
      import :: c_char, c_ptr, c_int
      type(c_ptr), value :: f
      integer(c_int), value :: varid
      character(kind=c_char), intent(in) :: name(*), text(*)
      integer(c_int) :: rc
    end function ncio_put_att_text_c
    function ncio_put_att_int_c(f, varid, name, vals, n) bind(C, name="ncio_put_att_int") result(rc)
      import :: c_char, c_ptr, c_int, c_int32_t
      type(c_ptr), value :: f
      integer(c_int), value :: varid, n
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int32_t), intent(in) :: vals(*)
      integer(c_int) :: rc
    end function ncio_put_att_int_c
    function ncio_reserve_start_c(path, nbytes) bind(C, name="ncio_reserve_start") result(rc)
      import :: c_char, c_int, c_int64_t
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int64_t), value :: nbytes
      integer(c_int) :: rc
    end function ncio_reserve_start_c
    function ncio_msleep(ms) bind(C, name="ncio_msleep") result(rc)
      import :: c_int
      integer(c_int), value :: ms
      integer(c_int) :: rc
    end function ncio_msleep
    function ncio_rename_c(from, to) bind(C, name="ncio_rename") result(rc)
      import :: c_char, c_int
      character(kind=c_char), intent(in) :: from(*), to(*)
      integer(c_int) :: rc
    end function ncio_rename_c
    function ncio_put_att_double_c(f, varid, name, vals, n) bind(C, name="ncio_put_att_double") result(rc)
      import :: c_char, c_ptr, c_int, c_double
      type(c_ptr), value :: f
      integer(c_int), value :: varid, n
      character(kind=c_char), intent(in) :: name(*)
      real(c_double), intent(in) :: vals(*)
      integer(c_int) :: rc
    end function ncio_put_att_double_c
    function ncio_get_att_text_c(f, varid, name, buf, buf_len) bind(C, name="ncio_get_att_text") result(rc)
      import :: c_char, c_ptr, c_int
      type(c_ptr), value :: f
      integer(c_int), value :: varid, buf_len
      character(kind=c_char), intent(in) :: name(*)
      character(kind=c_char), intent(out) :: buf(*)
      integer(c_int) :: rc
    end function ncio_get_att_text_c
    function ncio_put_att_float_c(f, varid, name, vals, n) bind(C, name="ncio_put_att_float") result(rc)
      import :: c_char, c_ptr, c_int, c_float
      type(c_ptr), value :: f
      integer(c_int), value :: varid, n
      character(kind=c_char), intent(in) :: name(*)
      real(c_float), intent(in) :: vals(*)
      integer(c_int) :: rc
    end function ncio_put_att_float_c
    function ncio_enddef(f) bind(C, name="ncio_enddef") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: f
      integer(c_int) :: rc
    end function ncio_enddef
    function ncio_put_var(f, varid, rec, mem_type, buf) bind(C, name="ncio_put_var") result(rc)
      import :: c_ptr, c_int, c_int64_t
      type(c_ptr), value :: f
      integer(c_int), value :: varid, mem_type
      integer(c_int64_t), value :: rec
      type(*), dimension(*), intent(in) :: buf
      integer(c_int) :: rc
    end function ncio_put_var
    function ncio_close(f) bind(C, name="ncio_close") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: f
      integer(c_int) :: rc
    end function ncio_close
  end interface

contains

  function cstr(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char, len=:), allocatable :: c
    c = trim(s)//c_null_char
  end function cstr

  function ncio_strerror() result(msg)
    character(len=:), allocatable :: msg
    character(kind=c_char), pointer :: p(:)
    type(c_ptr) :: cp
    integer :: n
    cp = ncio_strerror_c()
    msg = ""
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, p, [512])
    n = 0
    do while (n < 512)
      if (p(n + 1) == c_null_char) exit
      n = n + 1
      msg = msg//p(n)
    end do
  end function ncio_strerror

  !> netcdf_err (utils.F90:34-47): print and stop on a negative return code
  subroutine ncio_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    if (rc >= 0) return
    print *, "FATAL ERROR: ", trim(what), ": ", ncio_strerror()
    error stop 999
  end subroutine ncio_check

  integer(c_int) function ncio_open(path, f) result(rc)
    character(len=*), intent(in) :: path
    type(c_ptr), intent(out) :: f
    rc = ncio_open_c(cstr(path), f)
  end function ncio_open
  integer(c_int) function ncio_reserve_start(path, nbytes) result(rc)
    character(len=*), intent(in) :: path
    integer(c_int64_t), intent(in) :: nbytes
    rc = ncio_reserve_start_c(cstr(path), nbytes)
  end function ncio_reserve_start
  integer(c_int) function ncio_create(path, fmt, f) result(rc)
    character(len=*), intent(in) :: path
    integer, intent(in) :: fmt
    type(c_ptr), intent(out) :: f
    rc = ncio_create_c(cstr(path), int(fmt, c_int), f)
  end function ncio_create
  integer(c_int) function ncio_inq_dim(f, name, len) result(rc)
    type(c_ptr), intent(in) :: f
    character(len=*), intent(in) :: name
    integer(c_int64_t), intent(out) :: len
    integer(c_int) :: unl
    rc = ncio_inq_dim_c(f, cstr(name), len, unl)
  end function ncio_inq_dim
  !> first value of a numeric global attribute (nf90_get_att), converted to double
  integer(c_int) function ncio_get_gatt(f, name, val) result(rc)
    type(c_ptr), intent(in) :: f
    character(len=*), intent(in) :: name
    real(c_double), intent(out) :: val
    real(c_double) :: v(1)
    integer(c_int) :: n
    v = 0.0_c_double
    rc = ncio_get_att_double_c(f, NCIO_GLOBAL, cstr(name), v, 1_c_int, n)
    val = v(1)
  end function ncio_get_gatt
  integer(c_int) function ncio_inq_varid(f, name, varid) result(rc)
    type(c_ptr), intent(in) :: f
    character(len=*), intent(in) :: name
    integer(c_int), intent(out) :: varid
    rc = ncio_inq_varid_c(f, cstr(name), varid)
  end function ncio_inq_varid
  integer(c_int) function ncio_def_dim(f, name, len, dimid) result(rc)
    type(c_ptr), intent(in) :: f
    character(len=*), intent(in) :: name
    integer, intent(in) :: len
    integer(c_int), intent(out) :: dimid
    rc = ncio_def_dim_c(f, cstr(name), int(len, c_int64_t), dimid)
  end function ncio_def_dim
  integer(c_int) function ncio_def_var(f, name, xtype, dimids, varid) result(rc)
    type(c_ptr), intent(in) :: f
    character(len=*), intent(in) :: name
    integer(c_int), intent(in) :: xtype
    integer(c_int), intent(in) :: dimids(:)     ! slowest first (C order), i.e. reversed nf90 order
    integer(c_int), intent(out) :: varid
    rc = ncio_def_var_c(f, cstr(name), xtype, int(size(dimids), c_int), dimids, varid)
  end function ncio_def_var
  integer(c_int) function ncio_put_att_text(f, varid, name, text) result(rc)
    type(c_ptr), intent(in) :: f
    integer(c_int), intent(in) :: varid
    character(len=*), intent(in) :: name, text
    rc = ncio_put_att_text_c(f, varid, cstr(name), text//c_null_char)
  end function ncio_put_att_text
  integer(c_int) function ncio_put_att_int(f, varid, name, val) result(rc)
    type(c_ptr), intent(in) :: f
    integer(c_int), intent(in) :: varid
    character(len=*), intent(in) :: name
    integer, intent(in) :: val
    rc = ncio_put_att_int_c(f, varid, cstr(name), [int(val, c_int32_t)], 1_c_int)
  end function ncio_put_att_int
  integer(c_int) function ncio_put_att_real(f, varid, name, val) result(rc)
    type(c_ptr), intent(in) :: f
    integer(c_int), intent(in) :: varid
    character(len=*), intent(in) :: name
    real(c_double), intent(in) :: val
    ! the reference is built with -r8 (CMakeLists.txt:80-82): its `real` attributes are NF90_DOUBLE
    rc = ncio_put_att_double_c(f, varid, cstr(name), [val], 1_c_int)
  end function ncio_put_att_real
  integer(c_int) function ncio_rename(from, to) result(rc)
    character(len=*), intent(in) :: from, to
    rc = ncio_rename_c(cstr(from), cstr(to))
  end function ncio_rename
  !> text global attribute (nf90_get_att into a character variable); blank-padded, rc /= 0 when absent or not text
  integer(c_int) function ncio_get_gatt_text(f, name, text) result(rc)
    type(c_ptr), intent(in) :: f
    character(len=*), intent(in) :: name
    character(len=*), intent(out) :: text
    character(kind=c_char) :: buf(512)
    integer :: i
    text = ""
    buf = c_null_char
    rc = ncio_get_att_text_c(f, NCIO_GLOBAL, cstr(name), buf, 512_c_int)
    if (rc /= 0) return
    do i = 1, min(len(text), 511)
      if (buf(i) == c_null_char) exit
      text(i:i) = buf(i)
    end do
  end function ncio_get_gatt_text
end module ncio
