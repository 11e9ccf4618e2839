/* NetCDF classic (CDF-1 / CDF-2 / CDF-5) reader and writer -- see include/mpassit_ncio.h.
 *
 * Written from the published on-disk grammar (NetCDF User's Guide, "File Format Specification"; CDF-5: PnetCDF
 * "CDF-5 file format"); not derived from libnetcdf sources.  Layout:
 *   header  = magic numrecs dim_list gatt_list var_list
 *   magic   = 'C' 'D' 'F' VERSION(1|2|5)
 *   NON_NEG = 4 bytes (CDF-1/2) | 8 bytes (CDF-5); begin offsets are 4 bytes in CDF-1, 8 bytes in CDF-2/5
 *   lists   = ABSENT (ZERO NON_NEG(0)) | TAG(0x0A dim, 0x0B var, 0x0C att) NON_NEG(nelems) elements
 *   name    = NON_NEG(len) bytes padded to 4;  att = name type NON_NEG(n) values padded to 4
 *   var     = name NON_NEG(ndims) dimid* vatt_list type vsize begin     (dimid = NON_NEG)
 *   data    = non-record variables (each padded to 4), then numrecs records = one slab of every record variable
 * All numbers big-endian.  Host is assumed little-endian (x86-64, the only host of an MI355X node).           */
#define _FILE_OFFSET_BITS 64
#include "../../include/mpassit_ncio.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#define TAG_DIM 0x0A
#define TAG_VAR 0x0B
#define TAG_ATT 0x0C
#define CHUNK (1 << 18) /* elements converted per I/O chunk */

typedef struct { char *name; int64_t len; } dim_t;
typedef struct { char *name; int type; int64_t n; void *data; /* host order */ } att_t;
typedef struct {
  char *name;
  int ndims, dimids[NCIO_MAX_DIMS];
  int natts;
  att_t *atts;
  int type, is_rec;
  int64_t vsize, begin, count; /* count = elements per record (record var) or in total */
} var_t;
struct ncio_file {
  FILE *fp;
  int writing, defmode, format;
  int64_t numrecs, recsize;
  int ndims, nvars, ngatts, recdim;
  dim_t *dims;
  var_t *vars;
  att_t *gatts;
  int64_t data_end; /* writer: end of the non-record section / start of records */
  int64_t rec_start;
  void *h5; /* NetCDF-4: the HDF5 backend's state (nc4hdf5.h); the header above is shared, fp is NULL */
};

static __thread char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
const char *ncio_strerror(void) { return g_err; }

static int tsize(int t) {
  static const int s[12] = {0, 1, 1, 2, 4, 4, 8, 1, 2, 4, 8, 8};
  return (t >= 1 && t <= 11) ? s[t] : 0;
}
static int64_t pad4(int64_t n) { return (n + 3) & ~(int64_t)3; }
static uint16_t bs16(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }
static uint32_t bs32(uint32_t v) { return __builtin_bswap32(v); }
static uint64_t bs64(uint64_t v) { return __builtin_bswap64(v); }
static void swap_buf(void *p, int64_t n, int size) {
  if (size == 2) { uint16_t *q = (uint16_t *)p; for (int64_t i = 0; i < n; ++i) q[i] = bs16(q[i]); }
  else if (size == 4) { uint32_t *q = (uint32_t *)p; for (int64_t i = 0; i < n; ++i) q[i] = bs32(q[i]); }
  else if (size == 8) { uint64_t *q = (uint64_t *)p; for (int64_t i = 0; i < n; ++i) q[i] = bs64(q[i]); }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* header reading                                                                                               */
typedef struct { FILE *fp; int fmt; int bad; int64_t fsize; } rd_t;   /* fsize bounds every count the header claims */
static uint32_t rd_u32(rd_t *r) {
  uint32_t v = 0;
  if (fread(&v, 4, 1, r->fp) != 1) r->bad = 1;
  return bs32(v);
}
static int64_t rd_u64(rd_t *r) {
  uint64_t v = 0;
  if (fread(&v, 8, 1, r->fp) != 1) r->bad = 1;
  return (int64_t)bs64(v);
}
static int64_t rd_nonneg(rd_t *r) { return r->fmt == 5 ? rd_u64(r) : (int64_t)rd_u32(r); }
static char *rd_name(rd_t *r) {
  int64_t n = rd_nonneg(r);
  if (r->bad || n < 0 || n > 4096) { r->bad = 1; return NULL; }
  char *s = (char *)calloc((size_t)pad4(n) + 1, 1);
  if (!s) { r->bad = 1; return NULL; }
  if (pad4(n) && fread(s, 1, (size_t)pad4(n), r->fp) != (size_t)pad4(n)) r->bad = 1;
  s[n] = 0;
  return s;
}
static int rd_atts(rd_t *r, int *natts, att_t **atts) {
  uint32_t tag = rd_u32(r);
  int64_t n = rd_nonneg(r);
  *natts = 0;
  *atts = NULL;
  if (r->bad) return -1;
  if (tag == 0 && n == 0) return 0;
  if (tag != TAG_ATT || n < 0 || n > 100000) { r->bad = 1; return -1; }
  *atts = (att_t *)calloc((size_t)n, sizeof(att_t));
  if (!*atts) { r->bad = 1; return -1; }
  *natts = (int)n;
  for (int64_t a = 0; a < n && !r->bad; ++a) {
    att_t *t = &(*atts)[a];
    t->name = rd_name(r);
    t->type = (int)rd_u32(r);
    t->n = rd_nonneg(r);
    int sz = tsize(t->type);
    /* a damaged or hostile header must end in NCIO_EFORMAT, not in an overflowing product, a huge allocation or a NULL
     * dereference: an attribute cannot hold more values than the file has bytes */
    if (r->bad || !sz || t->n < 0 || t->n > r->fsize / sz) { r->bad = 1; t->n = 0; break; }
    int64_t bytes = pad4(t->n * sz);
    t->data = calloc((size_t)bytes + 1, 1);
    if (!t->data) { r->bad = 1; t->n = 0; break; }
    if (bytes && fread(t->data, 1, (size_t)bytes, r->fp) != (size_t)bytes) { r->bad = 1; t->n = 0; break; }
    swap_buf(t->data, t->n, sz);
  }
  return r->bad ? -1 : 0;
}

static void h5_free(ncio_file *f); /* nc4hdf5.h */
static void free_atts(int n, att_t *a) {
  for (int i = 0; i < n; ++i) { free(a[i].name); free(a[i].data); }
  free(a);
}
static void free_file(ncio_file *f) {
  if (!f) return;
  if (f->h5) h5_free(f);
  for (int i = 0; i < f->ndims; ++i) free(f->dims[i].name);
  free(f->dims);
  free_atts(f->ngatts, f->gatts);
  for (int i = 0; i < f->nvars; ++i) { free(f->vars[i].name); free_atts(f->vars[i].natts, f->vars[i].atts); }
  free(f->vars);
  if (f->fp) fclose(f->fp);
  free(f);
}

static void finish_layout_info(ncio_file *f) { /* per-variable element counts, record flags, record size */
  int nrec = 0;
  f->recsize = 0;
  for (int v = 0; v < f->nvars; ++v) {
    var_t *x = &f->vars[v];
    x->is_rec = x->ndims > 0 && x->dimids[0] == f->recdim;
    x->count = 1;
    for (int d = x->is_rec ? 1 : 0; d < x->ndims; ++d) x->count *= f->dims[x->dimids[d]].len;
    if (x->is_rec) { ++nrec; f->recsize += pad4(x->count * tsize(x->type)); } /* not the header's vsize: CDF-1/2 cap it */
  }
  if (nrec == 1) /* a lone record variable is stored without record padding */
    for (int v = 0; v < f->nvars; ++v)
      if (f->vars[v].is_rec) f->recsize = f->vars[v].count * tsize(f->vars[v].type);
}

static int convert_mt(int st, const void *src, int dt, void *dst, int64_t n); /* below */
#include "nc4hdf5.h"

int ncio_open(const char *path, ncio_file **out) {
  if (!path || !out) return fail(NCIO_EINVAL, "ncio_open: NULL argument");
  FILE *fp = fopen(path, "rb");
  if (!fp) return fail(NCIO_EIO, "ncio_open: cannot open %s", path);
  unsigned char m[8] = {0};
  if (fread(m, 1, 4, fp) != 4) { fclose(fp); return fail(NCIO_EFORMAT, "ncio_open: %s is too short to be a NetCDF file", path); }
  if (m[0] == 0x89 && m[1] == 'H' && m[2] == 'D' && m[3] == 'F') {
    fclose(fp);
    return nc4_open(path, out); /* NetCDF-4 = an HDF5 container: through libhdf5 where this build has it */
  }
  if (m[0] != 'C' || m[1] != 'D' || m[2] != 'F' || (m[3] != 1 && m[3] != 2 && m[3] != 5)) {
    fclose(fp);
    return fail(NCIO_EFORMAT, "ncio_open: %s is not a NetCDF classic file (magic %02x %02x %02x %02x)", path, m[0], m[1], m[2], m[3]);
  }
  ncio_file *f = (ncio_file *)calloc(1, sizeof(*f));
  if (!f) { fclose(fp); return fail(NCIO_ENOMEM, "out of memory"); }
  f->fp = fp;
  f->format = m[3];
  f->recdim = -1;
  rd_t r = {fp, f->format, 0, 0};
  if (fseeko(fp, 0, SEEK_END) == 0) r.fsize = (int64_t)ftello(fp);
  if (r.fsize <= 0 || fseeko(fp, 4, SEEK_SET) != 0) { free_file(f); return fail(NCIO_EIO, "ncio_open: cannot size %s", path); }
  f->numrecs = rd_nonneg(&r);
  if (f->format != 5 && f->numrecs == 0xFFFFFFFFll) f->numrecs = 0; /* STREAMING marker */
  uint32_t tag = rd_u32(&r);
  int64_t n = rd_nonneg(&r);
  if (!r.bad && !(tag == 0 && n == 0)) {
    if (tag != TAG_DIM || n < 0 || n > 100000 || n > r.fsize / 8) r.bad = 1;
    else if (!(f->dims = (dim_t *)calloc((size_t)n, sizeof(dim_t)))) r.bad = 1;
    else {
      f->ndims = (int)n;
      for (int d = 0; d < f->ndims && !r.bad; ++d) {
        f->dims[d].name = rd_name(&r);
        f->dims[d].len = rd_nonneg(&r);
        if (f->dims[d].len == 0) f->recdim = d;
      }
    }
  }
  if (!r.bad) rd_atts(&r, &f->ngatts, &f->gatts);
  if (!r.bad) {
    tag = rd_u32(&r);
    n = rd_nonneg(&r);
    if (!(tag == 0 && n == 0)) {
      if (tag != TAG_VAR || n < 0 || n > 1000000 || n > r.fsize / 16) r.bad = 1;
      else if (!(f->vars = (var_t *)calloc((size_t)n, sizeof(var_t)))) r.bad = 1;
      else {
        f->nvars = (int)n;
        for (int v = 0; v < f->nvars && !r.bad; ++v) {
          var_t *x = &f->vars[v];
          x->name = rd_name(&r);
          int64_t nd = rd_nonneg(&r);
          if (nd < 0 || nd > NCIO_MAX_DIMS) { r.bad = 1; break; }
          x->ndims = (int)nd;
          for (int d = 0; d < x->ndims; ++d) {
            int64_t id = rd_nonneg(&r);
            if (id < 0 || id >= f->ndims) r.bad = 1;
            x->dimids[d] = (int)id;
          }
          if (r.bad) break;
          rd_atts(&r, &x->natts, &x->atts);
          x->type = (int)rd_u32(&r);
          x->vsize = rd_nonneg(&r);
          x->begin = f->format == 1 ? (int64_t)rd_u32(&r) : rd_u64(&r);
          if (!tsize(x->type)) r.bad = 1;
        }
      }
    }
  }
  if (r.bad) { free_file(f); return fail(NCIO_EFORMAT, "ncio_open: %s has a damaged or truncated header", path); }
  if (f->recdim >= 0) f->dims[f->recdim].len = 0;
  /* every variable must lie inside the file: a corrupted dimension length or offset is caught here, not as an absurd
   * allocation or a short read later (the record count is checked against the record size the same way) */
  for (int v = 0; v < f->nvars; ++v) {
    var_t *x = &f->vars[v];
    int64_t n = 1;
    int isrec = 0, ok = x->begin >= 0 && x->begin <= r.fsize && x->name != NULL;
    for (int d = 0; d < x->ndims && ok; ++d) {
      if (x->dimids[d] == f->recdim) { isrec = 1; if (d != 0) ok = 0; continue; }
      int64_t len = f->dims[x->dimids[d]].len;
      if (len < 0 || (len > 0 && n > r.fsize / len)) ok = 0;
      else n *= len;
    }
    if (ok && n > (r.fsize - x->begin) / tsize(x->type) && !(isrec && f->numrecs == 0)) ok = 0;
    if (ok && isrec && f->numrecs > 0 && n > 0 && f->numrecs > r.fsize / (n * tsize(x->type)) + 1) ok = 0;
    if (!ok) { free_file(f); return fail(NCIO_EFORMAT, "ncio_open: %s: variable %d does not fit the file (damaged header)", path, v); }
  }
  for (int d = 0; d < f->ndims; ++d)
    if (!f->dims[d].name) { free_file(f); return fail(NCIO_EFORMAT, "ncio_open: %s has a damaged header", path); }
  finish_layout_info(f);
  *out = f;
  return 0;
}

int ncio_format(ncio_file *f) { return f ? f->format : NCIO_EINVAL; }
int ncio_has_netcdf4(void) { return NC4_AVAILABLE; }
int64_t ncio_numrecs(ncio_file *f) { return f ? f->numrecs : NCIO_EINVAL; }
int ncio_ndims(ncio_file *f) { return f ? f->ndims : NCIO_EINVAL; }
int ncio_nvars(ncio_file *f) { return f ? f->nvars : NCIO_EINVAL; }

static void copy_name(char *buf, int len, const char *s) {
  if (buf && len > 0) { strncpy(buf, s, (size_t)len - 1); buf[len - 1] = 0; }
}
int ncio_inq_dim_by_id(ncio_file *f, int dimid, char *name_buf, int buf_len, int64_t *len, int *is_unlimited) {
  if (!f || dimid < 0 || dimid >= f->ndims) return fail(NCIO_EINVAL, "ncio_inq_dim_by_id: bad dimension id %d", dimid);
  copy_name(name_buf, buf_len, f->dims[dimid].name);
  if (len) *len = dimid == f->recdim ? f->numrecs : f->dims[dimid].len;
  if (is_unlimited) *is_unlimited = dimid == f->recdim;
  return 0;
}
int ncio_inq_dim(ncio_file *f, const char *name, int64_t *len, int *is_unlimited) {
  if (!f || !name) return fail(NCIO_EINVAL, "ncio_inq_dim: NULL argument");
  for (int d = 0; d < f->ndims; ++d)
    if (!strcmp(f->dims[d].name, name)) return ncio_inq_dim_by_id(f, d, NULL, 0, len, is_unlimited);
  return fail(NCIO_ENOTFOUND, "dimension %s not found", name);
}
int ncio_inq_varid(ncio_file *f, const char *name, int *varid) {
  if (!f || !name || !varid) return fail(NCIO_EINVAL, "ncio_inq_varid: NULL argument");
  for (int v = 0; v < f->nvars; ++v)
    if (!strcmp(f->vars[v].name, name)) { *varid = v; return 0; }
  return fail(NCIO_ENOTFOUND, "variable %s not found", name);
}
int ncio_inq_var(ncio_file *f, int varid, char *name_buf, int buf_len, int *type, int *ndims, int64_t *shape, int *dimids, int *is_record) {
  if (!f || varid < 0 || varid >= f->nvars) return fail(NCIO_EINVAL, "ncio_inq_var: bad variable id %d", varid);
  var_t *x = &f->vars[varid];
  copy_name(name_buf, buf_len, x->name);
  if (type) *type = x->type;
  if (ndims) *ndims = x->ndims;
  for (int d = 0; d < x->ndims; ++d) {
    if (shape) shape[d] = x->dimids[d] == f->recdim ? f->numrecs : f->dims[x->dimids[d]].len;
    if (dimids) dimids[d] = x->dimids[d];
  }
  if (is_record) *is_record = x->is_rec;
  return 0;
}

/* ---- element conversion ------------------------------------------------------------------------------------- */
#define CONV_LOOP(ST, DT) { const ST *s_ = (const ST *)src; DT *d_ = (DT *)dst; for (int64_t i_ = 0; i_ < n; ++i_) d_[i_] = (DT)s_[i_]; } break
#define CONV_FROM(ST)                      \
  switch (dt) {                            \
    case NCIO_BYTE: CONV_LOOP(ST, int8_t); \
    case NCIO_CHAR: CONV_LOOP(ST, char);   \
    case NCIO_UBYTE: CONV_LOOP(ST, uint8_t); \
    case NCIO_SHORT: CONV_LOOP(ST, int16_t); \
    case NCIO_USHORT: CONV_LOOP(ST, uint16_t); \
    case NCIO_INT: CONV_LOOP(ST, int32_t); \
    case NCIO_UINT: CONV_LOOP(ST, uint32_t); \
    case NCIO_FLOAT: CONV_LOOP(ST, float); \
    case NCIO_DOUBLE: CONV_LOOP(ST, double); \
    case NCIO_INT64: CONV_LOOP(ST, int64_t); \
    case NCIO_UINT64: CONV_LOOP(ST, uint64_t); \
    default: return -1;                    \
  }
/* host-order src of type st -> host-order dst of type dt */
static int convert(int st, const void *src, int dt, void *dst, int64_t n) {
  if (st == dt) { memcpy(dst, src, (size_t)(n * tsize(st))); return 0; }
  switch (st) {
    case NCIO_BYTE: CONV_FROM(int8_t); break;
    case NCIO_CHAR: CONV_FROM(char); break;
    case NCIO_UBYTE: CONV_FROM(uint8_t); break;
    case NCIO_SHORT: CONV_FROM(int16_t); break;
    case NCIO_USHORT: CONV_FROM(uint16_t); break;
    case NCIO_INT: CONV_FROM(int32_t); break;
    case NCIO_UINT: CONV_FROM(uint32_t); break;
    case NCIO_FLOAT: CONV_FROM(float); break;
    case NCIO_DOUBLE: CONV_FROM(double); break;
    case NCIO_INT64: CONV_FROM(int64_t); break;
    case NCIO_UINT64: CONV_FROM(uint64_t); break;
    default: return -1;
  }
  return 0;
}

/* the same over a few threads for large arrays (one core converts 1-2 GB/s; a configuration-4 variable is 0.4-0.8 GB) */
typedef struct { int st, dt, rc; const char *src; char *dst; int64_t n; } conv_job;
static void *conv_worker(void *arg) {
  conv_job *j = (conv_job *)arg;
  j->rc = convert(j->st, j->src, j->dt, j->dst, j->n);
  return NULL;
}
static int convert_mt(int st, const void *src, int dt, void *dst, int64_t n) {
  int nthr = 1;
  if (n * (int64_t)tsize(st) >= ((int64_t)32 << 20)) {
    const char *e = getenv("NCIO_THREADS");
    long t = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    nthr = t > 8 ? 8 : (t < 1 ? 1 : (int)t);
  }
  if (nthr == 1) return convert(st, src, dt, dst, n);
  conv_job job[8];
  pthread_t th[8];
  int started[8] = {0}, rc = 0;
  const int64_t per = (n + nthr - 1) / nthr;
  for (int t = 0; t < nthr; ++t) {
    const int64_t a = t * per, b = a + per < n ? a + per : n;
    job[t].st = st; job[t].dt = dt; job[t].rc = 0; job[t].n = b > a ? b - a : 0;
    job[t].src = (const char *)src + a * tsize(st);
    job[t].dst = (char *)dst + a * tsize(dt);
    if (t == nthr - 1 || pthread_create(&th[t], NULL, conv_worker, &job[t]) != 0) conv_worker(&job[t]);   /* the caller takes the last slice (and any that no thread took) */
    else started[t] = 1;
  }
  for (int t = 0; t < nthr; ++t) {
    if (started[t]) pthread_join(th[t], NULL);
    if (job[t].rc) rc = job[t].rc;
  }
  return rc;
}

static int var_offset(ncio_file *f, var_t *x, int64_t rec, int64_t *off, const char *who) {
  if (x->is_rec) {
    if (rec < 0 || (!f->writing && rec >= f->numrecs)) return fail(NCIO_ERANGE, "%s: record %lld of %s out of range (%lld records)", who,
                                                                  (long long)rec, x->name, (long long)f->numrecs);
    *off = x->begin + rec * f->recsize;
  } else {
    *off = x->begin;
  }
  return 0;
}

/* ---- large variables: the chunks of one get / put are spread over a few threads (pread / pwrite on the descriptor,
 * one conversion buffer each) -- a single core converts and copies at 2-3 GB/s, a 3 M-cell x 55-level variable is
 * 0.66 GB and a history file holds fifteen of them. */
#define PAR_MIN_BYTES ((int64_t)32 << 20)
#define PAR_MAX_THREADS 8
typedef struct {
  int fd, writing, src_type, dst_type, fs, ms;
  int64_t off, count, next, err;
  char *buf;
} par_job;

static int par_threads(int64_t nbytes) {
  if (nbytes < PAR_MIN_BYTES) return 1;
  const char *e = getenv("NCIO_THREADS");
  long t = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
  if (t > PAR_MAX_THREADS) t = PAR_MAX_THREADS;
  return t < 1 ? 1 : (int)t;
}

static int xfer_full(int fd, int writing, char *p, int64_t n, int64_t off) {
  while (n > 0) {
    ssize_t k = writing ? pwrite(fd, p, (size_t)n, (off_t)off) : pread(fd, p, (size_t)n, (off_t)off);
    if (k <= 0) return -1;
    p += k; off += k; n -= k;
  }
  return 0;
}

static void *par_worker(void *arg) {
  par_job *j = (par_job *)arg;
  void *tmp = malloc((size_t)CHUNK * 8);
  for (;;) {
    int64_t done = __atomic_fetch_add(&j->next, (int64_t)CHUNK, __ATOMIC_RELAXED);
    if (done >= j->count || __atomic_load_n(&j->err, __ATOMIC_RELAXED)) break;
    int64_t n = j->count - done < CHUNK ? j->count - done : CHUNK;
    int bad = !tmp;
    if (!bad && !j->writing) {
      bad = xfer_full(j->fd, 0, (char *)tmp, n * j->fs, j->off + done * j->fs);
      if (!bad) {
        swap_buf(tmp, n, j->fs);
        bad = convert(j->src_type, tmp, j->dst_type, j->buf + done * j->ms, n) ? 2 : 0;
      }
    } else if (!bad) {
      bad = convert(j->src_type, j->buf + done * j->ms, j->dst_type, tmp, n) ? 2 : 0;
      if (!bad) {
        swap_buf(tmp, n, j->fs);
        bad = xfer_full(j->fd, 1, (char *)tmp, n * j->fs, j->off + done * j->fs);
      }
    }
    if (bad) __atomic_store_n(&j->err, (int64_t)(bad == 2 ? 2 : 1), __ATOMIC_RELAXED);
  }
  free(tmp);
  return NULL;
}

/* 0 = done, 1 = I/O error, 2 = unsupported conversion */
static int par_xfer(par_job *j, int nthr) {
  pthread_t th[PAR_MAX_THREADS];
  int started = 0;
  for (int t = 1; t < nthr; ++t)
    if (pthread_create(&th[started], NULL, par_worker, j) == 0) ++started;
  par_worker(j); /* the caller works too, and alone if no thread could be started */
  for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
  return (int)j->err;
}

int ncio_get_var(ncio_file *f, int varid, int64_t rec, int mem_type, void *buf) {
  if (!f || f->writing) return fail(NCIO_EMODE, "ncio_get_var: file not open for reading");
  if (varid < 0 || varid >= f->nvars || !buf || !tsize(mem_type)) return fail(NCIO_EINVAL, "ncio_get_var: bad argument");
  if (f->h5) return nc4_xfer(f, varid, rec, mem_type, buf, 0, "ncio_get_var");
  var_t *x = &f->vars[varid];
  int64_t off;
  int rc = var_offset(f, x, rec, &off, "ncio_get_var");
  if (rc) return rc;
  const int fs = tsize(x->type), ms = tsize(mem_type);
  const int nthr = par_threads(x->count * fs);
  if (nthr > 1) {
    par_job j = {fileno(f->fp), 0, x->type, mem_type, fs, ms, off, x->count, 0, 0, (char *)buf};
    int e = par_xfer(&j, nthr);
    if (e) return e == 2 ? fail(NCIO_EINVAL, "ncio_get_var: unsupported conversion") : fail(NCIO_EIO, "ncio_get_var: short read of %s", x->name);
    return 0;
  }
  if (fseeko(f->fp, (off_t)off, SEEK_SET)) return fail(NCIO_EIO, "ncio_get_var: seek failed for %s", x->name);
  void *tmp = malloc((size_t)CHUNK * 8);
  if (!tmp) return fail(NCIO_ENOMEM, "out of memory");
  for (int64_t done = 0; done < x->count; done += CHUNK) {
    int64_t n = x->count - done < CHUNK ? x->count - done : CHUNK;
    if (fread(tmp, (size_t)fs, (size_t)n, f->fp) != (size_t)n) { free(tmp); return fail(NCIO_EIO, "ncio_get_var: short read of %s", x->name); }
    swap_buf(tmp, n, fs);
    if (convert(x->type, tmp, mem_type, (char *)buf + done * ms, n)) { free(tmp); return fail(NCIO_EINVAL, "ncio_get_var: unsupported conversion"); }
  }
  free(tmp);
  return 0;
}

static att_t *find_att(ncio_file *f, int varid, const char *name) {
  int n = varid == NCIO_GLOBAL ? f->ngatts : f->vars[varid].natts;
  att_t *a = varid == NCIO_GLOBAL ? f->gatts : f->vars[varid].atts;
  for (int i = 0; i < n; ++i)
    if (!strcmp(a[i].name, name)) return &a[i];
  return NULL;
}
int ncio_natts(ncio_file *f, int varid) {
  if (!f || varid < NCIO_GLOBAL || varid >= f->nvars) return fail(NCIO_EINVAL, "ncio_natts: bad argument");
  return varid == NCIO_GLOBAL ? f->ngatts : f->vars[varid].natts;
}
int ncio_inq_att(ncio_file *f, int varid, int index, char *name_buf, int buf_len, int *type, int64_t *n) {
  if (!f || varid < NCIO_GLOBAL || varid >= f->nvars) return fail(NCIO_EINVAL, "ncio_inq_att: bad argument");
  const int na = varid == NCIO_GLOBAL ? f->ngatts : f->vars[varid].natts;
  att_t *a = varid == NCIO_GLOBAL ? f->gatts : f->vars[varid].atts;
  if (index < 0 || index >= na) return fail(NCIO_ENOTFOUND, "ncio_inq_att: attribute %d of %d", index, na);
  copy_name(name_buf, buf_len, a[index].name);
  if (type) *type = a[index].type;
  if (n) *n = a[index].n;
  return 0;
}
int ncio_get_att_text(ncio_file *f, int varid, const char *name, char *buf, int buf_len) {
  if (!f || !name || !buf || buf_len < 1 || varid < NCIO_GLOBAL || varid >= f->nvars) return fail(NCIO_EINVAL, "ncio_get_att_text: bad argument");
  att_t *a = find_att(f, varid, name);
  if (!a) return fail(NCIO_ENOTFOUND, "attribute %s not found", name);
  if (a->type != NCIO_CHAR) return fail(NCIO_EINVAL, "attribute %s is not text", name);
  int64_t n = a->n < buf_len - 1 ? a->n : buf_len - 1;
  memcpy(buf, a->data, (size_t)n);
  buf[n] = 0;
  return 0;
}
int ncio_get_att_double(ncio_file *f, int varid, const char *name, double *vals, int max_n, int *n) {
  if (!f || !name || varid < NCIO_GLOBAL || varid >= f->nvars) return fail(NCIO_EINVAL, "ncio_get_att_double: bad argument");
  att_t *a = find_att(f, varid, name);
  if (!a) return fail(NCIO_ENOTFOUND, "attribute %s not found", name);
  if (a->type == NCIO_CHAR) return fail(NCIO_EINVAL, "attribute %s is text", name);
  if (n) *n = (int)a->n;
  int64_t m = a->n < max_n ? a->n : max_n;
  if (vals && m > 0 && convert(a->type, a->data, NCIO_DOUBLE, vals, m)) return fail(NCIO_EINVAL, "attribute %s: unsupported type", name);
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* writing                                                                                                      */
/* ---- output space reserved ahead of time -------------------------------------------------------------------------
 * Writing a multi-GB file that does not exist yet is page allocation: on a memory-backed file system (the GPU box's
 * /dev/shm) 9 GB of pwrite()s into a fresh file take 1.6-2.9 s, into already allocated pages 1.1 s, and posix_fallocate of
 * the 9 GB 0.54 s (tools/shm_write_probe.c, profiles/r03_shm_probe.txt).  A host that knows (an upper bound of) its
 * output size early starts the allocation on a helper thread while it still reads its inputs; ncio_create of the same
 * path then keeps the file instead of truncating it, and ncio_close trims it to its exact size. */
static struct {
  pthread_t th;
  int active, rc;
  char path[4096];
  int64_t nbytes;
} g_res;
static void *reserve_main(void *arg) {
  (void)arg;
  int fd = open(g_res.path, O_CREAT | O_TRUNC | O_WRONLY, 0644);   /* truncated first: unwritten ranges must read as zeros */
  if (fd < 0) {
    g_res.rc = -1;
    return NULL;
  }
  g_res.rc = posix_fallocate(fd, 0, (off_t)g_res.nbytes);
  close(fd);
  return NULL;
}
/* A process that ends before its writer has claimed the reservation (a host that stops on an error in between: the Fortran driver's
 * `error stop` runs the exit handlers) must not leave a file of zeros under the output's name. */
static void reserve_abandon(void) {
  if (!g_res.active) return;
  pthread_join(g_res.th, NULL);
  g_res.active = 0;
  unlink(g_res.path);
}
int ncio_reserve_start(const char *path, int64_t nbytes) {
  static int at_exit_set = 0;
  if (!path || nbytes <= 0 || strlen(path) >= sizeof(g_res.path)) return fail(NCIO_EINVAL, "ncio_reserve_start: bad argument");
  if (g_res.active) return fail(NCIO_EMODE, "ncio_reserve_start: a reservation is already running");
  if (!at_exit_set) {
    at_exit_set = 1;
    atexit(reserve_abandon);
  }
  strcpy(g_res.path, path);
  g_res.nbytes = nbytes;
  g_res.rc = 0;
  if (pthread_create(&g_res.th, NULL, reserve_main, NULL)) return fail(NCIO_EIO, "ncio_reserve_start: cannot start the helper thread");
  g_res.active = 1;
  return 0;
}
/* -> 1 when `path` was reserved by ncio_reserve_start and the allocation succeeded (the helper thread is joined) */
static int reserved_for(const char *path) {
  if (!g_res.active || strcmp(path, g_res.path)) return 0;
  pthread_join(g_res.th, NULL);
  g_res.active = 0;
  return g_res.rc == 0;
}

int ncio_create(const char *path, int format, ncio_file **out) {
  if (!path || !out || (format != 1 && format != 2 && format != 5 && format != 4)) return fail(NCIO_EINVAL, "ncio_create: format must be 1, 2, 5 or 4");
  if (format == 4) { /* NetCDF-4: libhdf5 creates the file (a pending reservation of the path is waited for, then overwritten) */
    (void)reserved_for(path);
    ncio_file *f4 = (ncio_file *)calloc(1, sizeof(*f4));
    if (!f4) return fail(NCIO_ENOMEM, "out of memory");
    f4->writing = f4->defmode = 1;
    f4->format = 4;
    f4->recdim = -1;
    int rc4 = nc4_create(path, f4);
    if (rc4) { free(f4); return rc4; }
    *out = f4;
    return 0;
  }
  /* a file this process reserved a moment ago (truncated, then allocated: all zeros) is kept; anything else is truncated */
  FILE *fp = fopen(path, reserved_for(path) ? "rb+" : "wb+");
  if (!fp) return fail(NCIO_EIO, "ncio_create: cannot create %s", path);
  ncio_file *f = (ncio_file *)calloc(1, sizeof(*f));
  if (!f) { fclose(fp); return fail(NCIO_ENOMEM, "out of memory"); }
  f->fp = fp;
  f->writing = f->defmode = 1;
  f->format = format;
  f->recdim = -1;
  *out = f;
  return 0;
}
static int need_def(ncio_file *f, const char *who) {
  if (!f || !f->writing || !f->defmode) return fail(NCIO_EMODE, "%s: file is not in define mode", who);
  return 0;
}
int ncio_def_dim(ncio_file *f, const char *name, int64_t len, int *dimid) {
  int rc = need_def(f, "ncio_def_dim");
  if (rc) return rc;
  if (!name || len < 0) return fail(NCIO_EINVAL, "ncio_def_dim: bad argument");
  if (len == 0 && f->recdim >= 0) return fail(NCIO_EINVAL, "ncio_def_dim: only one unlimited dimension per file");
  if (f->format != 5 && f->format != 4 && len > 0xFFFFFFFFll) return fail(NCIO_ERANGE, "ncio_def_dim: %s too long for CDF-%d", name, f->format);
  f->dims = (dim_t *)realloc(f->dims, sizeof(dim_t) * (size_t)(f->ndims + 1));
  f->dims[f->ndims].name = strdup(name);
  f->dims[f->ndims].len = len;
  if (len == 0) f->recdim = f->ndims;
  if (dimid) *dimid = f->ndims;
  f->ndims++;
  return 0;
}
int ncio_def_var(ncio_file *f, const char *name, int type, int ndims, const int *dimids, int *varid) {
  int rc = need_def(f, "ncio_def_var");
  if (rc) return rc;
  if (!name || !tsize(type) || ndims < 0 || ndims > NCIO_MAX_DIMS || (ndims && !dimids)) return fail(NCIO_EINVAL, "ncio_def_var: bad argument");
  if (f->format != 5 && f->format != 4 && type > NCIO_DOUBLE) return fail(NCIO_EINVAL, "ncio_def_var: type %d needs CDF-5 or NetCDF-4", type);
  f->vars = (var_t *)realloc(f->vars, sizeof(var_t) * (size_t)(f->nvars + 1));
  var_t *x = &f->vars[f->nvars];
  memset(x, 0, sizeof(*x));
  x->name = strdup(name);
  x->type = type;
  x->ndims = ndims;
  for (int d = 0; d < ndims; ++d) {
    if (dimids[d] < 0 || dimids[d] >= f->ndims) return fail(NCIO_EINVAL, "ncio_def_var: %s uses an undefined dimension", name);
    if (d > 0 && dimids[d] == f->recdim) return fail(NCIO_EINVAL, "ncio_def_var: the unlimited dimension must come first (%s)", name);
    x->dimids[d] = dimids[d];
  }
  if (varid) *varid = f->nvars;
  f->nvars++;
  return 0;
}
static int put_att(ncio_file *f, int varid, const char *name, int type, const void *vals, int64_t n) {
  int rc = need_def(f, "ncio_put_att");
  if (rc) return rc;
  if (!name || varid < NCIO_GLOBAL || varid >= f->nvars || n < 0 || (n && !vals)) return fail(NCIO_EINVAL, "ncio_put_att: bad argument");
  int *cnt = varid == NCIO_GLOBAL ? &f->ngatts : &f->vars[varid].natts;
  att_t **arr = varid == NCIO_GLOBAL ? &f->gatts : &f->vars[varid].atts;
  att_t *a = find_att(f, varid, name);
  if (a) free(a->data);
  else {
    *arr = (att_t *)realloc(*arr, sizeof(att_t) * (size_t)(*cnt + 1));
    a = &(*arr)[*cnt];
    a->name = strdup(name);
    (*cnt)++;
  }
  a->type = type;
  a->n = n;
  a->data = calloc((size_t)pad4(n * tsize(type)) + 1, 1);
  if (n) memcpy(a->data, vals, (size_t)(n * tsize(type)));
  return 0;
}
int ncio_put_att_text(ncio_file *f, int varid, const char *name, const char *text) {
  return put_att(f, varid, name, NCIO_CHAR, text ? text : "", text ? (int64_t)strlen(text) : 0);
}
int ncio_put_att_int(ncio_file *f, int varid, const char *name, const int32_t *vals, int n) { return put_att(f, varid, name, NCIO_INT, vals, n); }
int ncio_put_att_float(ncio_file *f, int varid, const char *name, const float *vals, int n) { return put_att(f, varid, name, NCIO_FLOAT, vals, n); }
int ncio_put_att_double(ncio_file *f, int varid, const char *name, const double *vals, int n) { return put_att(f, varid, name, NCIO_DOUBLE, vals, n); }

/* header serialisation into a growable buffer */
typedef struct { unsigned char *p; size_t n, cap; int fmt; } wb_t;
static void wb_raw(wb_t *w, const void *src, size_t n) {
  if (w->n + n > w->cap) { w->cap = (w->n + n) * 2 + 1024; w->p = (unsigned char *)realloc(w->p, w->cap); }
  memcpy(w->p + w->n, src, n);
  w->n += n;
}
static void wb_u32(wb_t *w, uint32_t v) { v = bs32(v); wb_raw(w, &v, 4); }
static void wb_u64(wb_t *w, uint64_t v) { v = bs64(v); wb_raw(w, &v, 8); }
static void wb_nonneg(wb_t *w, int64_t v) { if (w->fmt == 5) wb_u64(w, (uint64_t)v); else wb_u32(w, (uint32_t)v); }
static void wb_name(wb_t *w, const char *s) {
  size_t n = strlen(s);
  static const char zero[4] = {0, 0, 0, 0};
  wb_nonneg(w, (int64_t)n);
  wb_raw(w, s, n);
  wb_raw(w, zero, (size_t)(pad4((int64_t)n) - (int64_t)n));
}
static void wb_atts(wb_t *w, int n, att_t *a) {
  if (n == 0) { wb_u32(w, 0); wb_nonneg(w, 0); return; }
  wb_u32(w, TAG_ATT);
  wb_nonneg(w, n);
  for (int i = 0; i < n; ++i) {
    wb_name(w, a[i].name);
    wb_u32(w, (uint32_t)a[i].type);
    wb_nonneg(w, a[i].n);
    int sz = tsize(a[i].type);
    int64_t bytes = pad4(a[i].n * sz);
    unsigned char *tmp = (unsigned char *)calloc((size_t)bytes + 1, 1);
    memcpy(tmp, a[i].data, (size_t)(a[i].n * sz));
    swap_buf(tmp, a[i].n, sz);
    wb_raw(w, tmp, (size_t)bytes);
    free(tmp);
  }
}
static void serialise(ncio_file *f, wb_t *w) {
  unsigned char magic[4] = {'C', 'D', 'F', (unsigned char)f->format};
  w->n = 0;
  wb_raw(w, magic, 4);
  wb_nonneg(w, f->numrecs);
  if (f->ndims == 0) { wb_u32(w, 0); wb_nonneg(w, 0); }
  else {
    wb_u32(w, TAG_DIM);
    wb_nonneg(w, f->ndims);
    for (int d = 0; d < f->ndims; ++d) { wb_name(w, f->dims[d].name); wb_nonneg(w, f->dims[d].len); }
  }
  wb_atts(w, f->ngatts, f->gatts);
  if (f->nvars == 0) { wb_u32(w, 0); wb_nonneg(w, 0); }
  else {
    wb_u32(w, TAG_VAR);
    wb_nonneg(w, f->nvars);
    for (int v = 0; v < f->nvars; ++v) {
      var_t *x = &f->vars[v];
      wb_name(w, x->name);
      wb_nonneg(w, x->ndims);
      for (int d = 0; d < x->ndims; ++d) wb_nonneg(w, x->dimids[d]);
      wb_atts(w, x->natts, x->atts);
      wb_u32(w, (uint32_t)x->type);
      int64_t vs = x->vsize;
      if (f->format != 5 && vs > 0xFFFFFFFCll) vs = 0xFFFFFFFFll; /* "too big to represent" marker of CDF-1/2 */
      wb_nonneg(w, vs);
      if (f->format == 1) wb_u32(w, (uint32_t)x->begin);
      else wb_u64(w, (uint64_t)x->begin);
    }
  }
}

int ncio_enddef(ncio_file *f) {
  int rc = need_def(f, "ncio_enddef");
  if (rc) return rc;
  if (f->format == 4) return nc4_enddef(f);
  int nrec = 0;
  for (int v = 0; v < f->nvars; ++v) {
    var_t *x = &f->vars[v];
    x->is_rec = x->ndims > 0 && x->dimids[0] == f->recdim;
    x->count = 1;
    for (int d = x->is_rec ? 1 : 0; d < x->ndims; ++d) x->count *= f->dims[x->dimids[d]].len;
    x->vsize = pad4(x->count * tsize(x->type));
    nrec += x->is_rec;
  }
  wb_t w = {NULL, 0, 0, f->format};
  serialise(f, &w); /* sizes do not depend on the begin values */
  int64_t off = pad4((int64_t)w.n);
  for (int v = 0; v < f->nvars; ++v)
    if (!f->vars[v].is_rec) { f->vars[v].begin = off; off += f->vars[v].vsize; }
  f->rec_start = off;
  f->recsize = 0;
  for (int v = 0; v < f->nvars; ++v)
    if (f->vars[v].is_rec) { f->vars[v].begin = off; off += f->vars[v].vsize; f->recsize += f->vars[v].vsize; }
  if (nrec == 1) { /* keep records 4-byte aligned: declare the lone record variable's padded size as the record size */
    for (int v = 0; v < f->nvars; ++v)
      if (f->vars[v].is_rec && (f->vars[v].count * tsize(f->vars[v].type)) % 4) {
        free(w.p);
        return fail(NCIO_EINVAL, "ncio_enddef: a single record variable of %d-byte elements with an odd record size is not supported; "
                                 "add a second record variable", tsize(f->vars[v].type));
      }
  }
  if (f->format == 1 && off > 0x7FFFFFFFll) { free(w.p); return fail(NCIO_ERANGE, "ncio_enddef: layout exceeds CDF-1 offsets; use format 2 or 5"); }
  if (f->format != 5)
    for (int v = 0; v < f->nvars; ++v)
      if (f->vars[v].vsize > 0xFFFFFFFCll && !(v == f->nvars - 1 || f->vars[v].is_rec))
        { free(w.p); return fail(NCIO_ERANGE, "ncio_enddef: variable %s exceeds 4 GiB; use format 5", f->vars[v].name); }
  serialise(f, &w);
  if (fseeko(f->fp, 0, SEEK_SET) || fwrite(w.p, 1, w.n, f->fp) != w.n) { free(w.p); return fail(NCIO_EIO, "ncio_enddef: header write failed"); }
  free(w.p);
  f->defmode = 0;
  f->data_end = f->rec_start;
  return 0;
}

int ncio_put_var(ncio_file *f, int varid, int64_t rec, int mem_type, const void *buf) {
  if (!f || !f->writing || f->defmode) return fail(NCIO_EMODE, "ncio_put_var: call ncio_enddef first");
  if (varid < 0 || varid >= f->nvars || !buf || !tsize(mem_type)) return fail(NCIO_EINVAL, "ncio_put_var: bad argument");
  if (f->h5) return nc4_xfer(f, varid, rec, mem_type, (void *)(uintptr_t)buf, 1, "ncio_put_var");
  var_t *x = &f->vars[varid];
  int64_t off;
  int rc = var_offset(f, x, rec, &off, "ncio_put_var");
  if (rc) return rc;
  const int fs = tsize(x->type), ms = tsize(mem_type);
  const int nthr = par_threads(x->count * fs);
  if (nthr > 1) {
    if (fflush(f->fp)) return fail(NCIO_EIO, "ncio_put_var: flush failed");
    par_job j = {fileno(f->fp), 1, mem_type, x->type, fs, ms, off, x->count, 0, 0, (char *)(uintptr_t)buf};
    int e = par_xfer(&j, nthr);
    if (e) return e == 2 ? fail(NCIO_EINVAL, "ncio_put_var: unsupported conversion") : fail(NCIO_EIO, "ncio_put_var: short write of %s", x->name);
    if (x->is_rec && rec + 1 > f->numrecs) f->numrecs = rec + 1;
    return 0;
  }
  if (fseeko(f->fp, (off_t)off, SEEK_SET)) return fail(NCIO_EIO, "ncio_put_var: seek failed for %s", x->name);
  void *tmp = malloc((size_t)CHUNK * 8);
  if (!tmp) return fail(NCIO_ENOMEM, "out of memory");
  for (int64_t done = 0; done < x->count; done += CHUNK) {
    int64_t n = x->count - done < CHUNK ? x->count - done : CHUNK;
    if (convert(mem_type, (const char *)buf + done * ms, x->type, tmp, n)) { free(tmp); return fail(NCIO_EINVAL, "ncio_put_var: unsupported conversion"); }
    swap_buf(tmp, n, fs);
    if (fwrite(tmp, (size_t)fs, (size_t)n, f->fp) != (size_t)n) { free(tmp); return fail(NCIO_EIO, "ncio_put_var: short write of %s", x->name); }
  }
  free(tmp);
  if (x->is_rec && rec + 1 > f->numrecs) f->numrecs = rec + 1;
  return 0;
}

int ncio_var_extent(ncio_file *f, int varid, int64_t rec, int64_t *offset, int64_t *nbytes) {
  if (!f || varid < 0 || varid >= f->nvars) return fail(NCIO_EINVAL, "ncio_var_extent: bad argument");
  if (f->h5) return fail(NCIO_EMODE, "ncio_var_extent: a NetCDF-4 file has no raw byte range to offer (chunked / compressed / little-endian container): "
                                    "use ncio_get_var / ncio_put_var");
  if (f->writing && f->defmode) return fail(NCIO_EMODE, "ncio_var_extent: call ncio_enddef first");
  var_t *x = &f->vars[varid];
  int64_t off;
  int rc = var_offset(f, x, rec, &off, "ncio_var_extent");
  if (rc) return rc;
  const int64_t nb = x->count * tsize(x->type);
  if (f->writing) { /* make the range exist so that it can be mapped and written in place */
    if (x->is_rec && rec + 1 > f->numrecs) f->numrecs = rec + 1;
    int64_t end = f->rec_start + f->numrecs * f->recsize;
    if (end < off + nb) end = off + nb;
    fflush(f->fp);
    off_t cur = lseek(fileno(f->fp), 0, SEEK_END);
    if (cur < (off_t)end && ftruncate(fileno(f->fp), (off_t)end)) return fail(NCIO_EIO, "ncio_var_extent: cannot extend the file");
  }
  if (offset) *offset = off;
  if (nbytes) *nbytes = nb;
  return 0;
}

int ncio_close(ncio_file *f) {
  if (!f) return 0;
  int rc = 0;
  if (f->format == 4) {
    if (f->writing && f->defmode) rc = nc4_enddef(f);
    int rc2 = nc4_close(f);
    free_file(f);
    return rc ? rc : rc2;
  }
  if (f->writing) {
    if (f->defmode) rc = ncio_enddef(f);
    if (!rc) {
      /* numrecs sits right after the magic; unwritten space reads as zeros */
      wb_t w = {NULL, 0, 0, f->format};
      wb_nonneg(&w, f->numrecs);
      if (fseeko(f->fp, 4, SEEK_SET) || fwrite(w.p, 1, w.n, f->fp) != w.n) rc = fail(NCIO_EIO, "ncio_close: cannot update numrecs");
      free(w.p);
      int64_t end = f->rec_start + f->numrecs * f->recsize;
      fflush(f->fp);
      if (!rc && ftruncate(fileno(f->fp), (off_t)end)) rc = fail(NCIO_EIO, "ncio_close: cannot size the file");
    }
  }
  free_file(f);
  return rc;
}

/* ---- two POSIX helpers of the multi-image Fortran driver (ncfiles_mod.F90: marker files between driver images) ---- */
int ncio_msleep(int ms) {
  if (ms > 0) usleep((useconds_t)ms * 1000u);
  return 0;
}
int ncio_rename(const char *from, const char *to) {
  if (!from || !to) return fail(NCIO_EINVAL, "ncio_rename: NULL argument");
  return rename(from, to) == 0 ? 0 : fail(NCIO_EIO, "ncio_rename: cannot rename %s to %s", from, to);
}
