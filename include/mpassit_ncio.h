/* mpassit_ncio.h -- NetCDF CLASSIC container I/O for the host side of the hot path (no libnetcdf in the build image).
 *
 * Stands in for the nf90_* calls on either side of the interpolation:
 *   input   nf90_open / nf90_inq_dimid / nf90_inquire_dimension / nf90_inq_varid / nf90_get_var
 *           (model_grid.F90:287-417 grid file, input_data.F90:145-812 diag / hist files)
 *   output  nf90_create / nf90_def_dim / nf90_def_var / nf90_put_att / nf90_enddef / nf90_put_var / nf90_close
 *           (write_data.F90:173-1498)
 * Formats: CDF-1 (classic), CDF-2 (64-bit offset), CDF-5 (64-bit data) -- the on-disk layout published in the
 * NetCDF User's Guide ("File Format Specification") and the PnetCDF CDF-5 note; big-endian data, 4-byte padding,
 * record variables interleaved per record.  These are read and written by code of this repository alone.
 * NetCDF-4 files are HDF5 containers (the reference creates its output with NF90_NETCDF4, write_data.F90:173): where the build finds
 * the HDF5 C library (hdf5.h, libhdf5, libhdf5_hl: the reference's own dependency under libnetcdf; mpassit_amd/build.py find_hdf5) the
 * same calls read them (ncio_open recognises the magic) and write them (ncio_create with format 4) following the "NetCDF-4 File
 * Format" appendix of the User's Guide -- hostio/nc4hdf5.h; such files offer no raw byte range (ncio_var_extent).  A build without
 * HDF5 reports NCIO_EHDF5 for them and says how to convert.  One difference from a libnetcdf-written file: this writer sets no HDF5 fill value
 * and writes no _FillValue attribute, so a record that was never written reads back as 0, not as libnetcdf's default fill (9.96921e+36 for
 * NC_FLOAT) -- the driver writes every record of every variable it defines.  Checked against h5py, not against libnetcdf itself (absent here).
 * Host code, no GPU involved.
 * All functions return 0 on success or a negative NCIO_E* code; ncio_strerror() explains the last failure. */
#ifndef MPASSIT_NCIO_H
#define MPASSIT_NCIO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ncio_file ncio_file;

enum { NCIO_BYTE = 1, NCIO_CHAR = 2, NCIO_SHORT = 3, NCIO_INT = 4, NCIO_FLOAT = 5, NCIO_DOUBLE = 6,
       NCIO_UBYTE = 7, NCIO_USHORT = 8, NCIO_UINT = 9, NCIO_INT64 = 10, NCIO_UINT64 = 11 };
enum { NCIO_EIO = -1, NCIO_EFORMAT = -2, NCIO_EHDF5 = -3, NCIO_ENOTFOUND = -4, NCIO_EINVAL = -5, NCIO_ERANGE = -6,
       NCIO_EMODE = -7, NCIO_ENOMEM = -8 };
#define NCIO_MAX_DIMS 8
#define NCIO_GLOBAL (-1)

const char *ncio_strerror(void);

/* ---- reading ------------------------------------------------------------------------------------------------ */
int ncio_open(const char *path, ncio_file **out);
int ncio_format(ncio_file *f);                  /* 1, 2, 5 -- or 4: a NetCDF-4 file (HDF5 container) */
int ncio_has_netcdf4(void);                     /* 1 when this build has the HDF5 backend (see the head of this file), 0 otherwise */
int64_t ncio_numrecs(ncio_file *f);
int ncio_ndims(ncio_file *f);
int ncio_nvars(ncio_file *f);
int ncio_inq_dim(ncio_file *f, const char *name, int64_t *len, int *is_unlimited);
int ncio_inq_dim_by_id(ncio_file *f, int dimid, char *name_buf, int buf_len, int64_t *len, int *is_unlimited);
int ncio_inq_varid(ncio_file *f, const char *name, int *varid);
/* shape: dimension lengths, slowest first as stored (the record dimension reports the current number of records) */
int ncio_inq_var(ncio_file *f, int varid, char *name_buf, int buf_len, int *type, int *ndims, int64_t *shape, int *dimids,
                 int *is_record);
/* One whole non-record variable, or record `rec` (0-based) of a record variable, converted to mem_type
 * (NCIO_INT, NCIO_FLOAT, NCIO_DOUBLE, NCIO_INT64, or the variable's own type for BYTE/CHAR/SHORT) in host byte order.
 * buf must hold the variable's (record's) element count. */
int ncio_get_var(ncio_file *f, int varid, int64_t rec, int mem_type, void *buf);
/* attributes: varid = NCIO_GLOBAL for global ones.  Text: copied NUL-terminated (truncated to buf_len-1).
 * Numeric: converted to double, up to max_n values; *n receives the stored count. */
int ncio_natts(ncio_file *f, int varid);        /* number of attributes of a variable (NCIO_GLOBAL: of the file) */
int ncio_inq_att(ncio_file *f, int varid, int index, char *name_buf, int buf_len, int *type, int64_t *n);   /* nf90_inq_attname + nf90_inquire_attribute */
int ncio_get_att_text(ncio_file *f, int varid, const char *name, char *buf, int buf_len);
int ncio_get_att_double(ncio_file *f, int varid, const char *name, double *vals, int max_n, int *n);

/* ---- writing ------------------------------------------------------------------------------------------------ */
int ncio_create(const char *path, int format /* 1, 2, 5; 4 = NetCDF-4 (needs the HDF5 backend) */, ncio_file **out);
int ncio_def_dim(ncio_file *f, const char *name, int64_t len /* 0 = unlimited (one per file) */, int *dimid);
int ncio_def_var(ncio_file *f, const char *name, int type, int ndims, const int *dimids, int *varid);
int ncio_put_att_text(ncio_file *f, int varid, const char *name, const char *text);
int ncio_put_att_int(ncio_file *f, int varid, const char *name, const int32_t *vals, int n);
int ncio_put_att_float(ncio_file *f, int varid, const char *name, const float *vals, int n);
int ncio_put_att_double(ncio_file *f, int varid, const char *name, const double *vals, int n);
int ncio_enddef(ncio_file *f);                  /* lays the file out; data calls only after this */
/* whole variable / one record from host memory of mem_type (converted to the variable's type and byte-swapped) */
int ncio_put_var(ncio_file *f, int varid, int64_t rec, int mem_type, const void *buf);

/* ---- raw access (device-side ingest / egress) ---------------------------------------------------------------------
 * Byte range of a whole non-record variable, or of record `rec` of a record variable, inside the file: big-endian
 * elements of the variable's type, exactly as stored.  A caller can mmap / pread that range, move it to the GPU
 * untouched and swap bytes there (mpg_bswap_dev), or write GPU-produced big-endian bytes into it.  On a file being
 * written (after ncio_enddef) the range of record `rec` is made to exist (the file is extended, numrecs follows). */
int ncio_var_extent(ncio_file *f, int varid, int64_t rec, int64_t *offset, int64_t *nbytes);

int ncio_close(ncio_file *f);                   /* writer: fills numrecs, flushes, sizes the file exactly */

/* Start allocating `nbytes` (an upper bound of the output size) for a file that ncio_create will open shortly: the file is
 * created / truncated and its pages are allocated on a helper thread while the caller goes on (e.g. reads its inputs);
 * ncio_create of the same path waits for the helper and keeps the file, ncio_close trims it.  Unwritten ranges read as
 * zeros either way.  One reservation at a time.  A process that exits without having claimed its reservation (it stopped on an error
 * in between) takes the file away again: no file of zeros is left under the output's name. */
int ncio_reserve_start(const char *path, int64_t nbytes);

/* two POSIX helpers for hosts that coordinate several driver images through marker files (sleep; atomic rename) */
int ncio_msleep(int milliseconds);
int ncio_rename(const char *from, const char *to);

#ifdef __cplusplus
}
#endif
#endif
