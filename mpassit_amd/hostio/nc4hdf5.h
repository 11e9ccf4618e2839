/* NetCDF-4 (HDF5 container) backend of the ncio interface -- included by ncclassic.c, which holds the in-memory header
 * (dims / vars / attributes) both backends share.
 *
 * The reference reads whatever nf90_open accepts and creates its output with NF90_NETCDF4 (write_data.F90:173).  A NetCDF-4 file is
 * an HDF5 file that follows the conventions of the NetCDF User's Guide, appendix "NetCDF-4 File Format": every dimension is an HDF5
 * DIMENSION SCALE -- the coordinate variable's dataset, or a dataset of its own whose NAME attribute starts with "This is a netCDF
 * dimension but not a netCDF variable." --, every variable a dataset with its scales attached per axis (DIMENSION_LIST /
 * REFERENCE_LIST, maintained by the H5DS calls of libhdf5_hl), `_Netcdf4Dimid` numbers the dimensions, variables along an unlimited
 * dimension are chunked, link / attribute creation order is tracked so that the definition order survives, text attributes are
 * fixed-length scalar strings, NC_CHAR variables are datasets of 1-byte strings.
 *
 * This backend goes through the HDF5 C library (the reference's own dependency, under libnetcdf): compiled in when build.py finds
 * hdf5.h + libhdf5 + libhdf5_hl (MPASSIT_HDF5_ROOT, or the usual prefixes; the image has 1.10.6 under /opt/conda), a stub that says so
 * otherwise.  Reading: the header is copied into the shared in-memory form at open, ncio_get_var = H5Dread (whole variable, or one
 * record as a hyperslab; type conversion by the library; chunked, compressed, big- or little-endian data alike).  Writing: objects
 * are created at ncio_enddef from the definitions, ncio_put_var = H5Dwrite.  There is no raw byte range (ncio_var_extent refuses):
 * the hosts take their host-array flow for such files.  Files without dimension scales (plain HDF5) are read with anonymous
 * dimensions "phony_dim_N", as libnetcdf does.  Not handled: groups below the root, user-defined / variable-length / string-array
 * types (such variables are skipped), more than one unlimited dimension (the first is the record dimension, the others read at their
 * current length).
 * Checked against an independent implementation both ways (tests/test_nc4.py, h5py on libhdf5 1.10.6): files written here carry the
 * scales, lists and attributes h5py expects and the data it reads back; files h5py wrote with netCDF-4's conventions (chunked,
 * deflated, big-endian variants included) are read here.  NOT checked against libnetcdf itself (absent from the image). */
#ifdef MPASSIT_HAVE_HDF5
#include <hdf5.h>
#include <hdf5_hl.h>

#define NC4_DIM_WITHOUT_VAR "This is a netCDF dimension but not a netCDF variable."
#define NC4_CHUNK_TARGET ((int64_t)8 << 20)

typedef struct {
  hid_t file;
  hid_t *dset;   /* per variable */
  hid_t *scale;  /* per dimension: its dimension scale -- a dataset of its own (scale_own = 1) or the coordinate variable's */
  int *scale_own;
  int nscale;
} h5_t;

static hid_t h5_memtype(int t) {
  switch (t) {
    case NCIO_BYTE: return H5T_NATIVE_INT8;
    case NCIO_UBYTE: return H5T_NATIVE_UINT8;
    case NCIO_SHORT: return H5T_NATIVE_INT16;
    case NCIO_USHORT: return H5T_NATIVE_UINT16;
    case NCIO_INT: return H5T_NATIVE_INT32;
    case NCIO_UINT: return H5T_NATIVE_UINT32;
    case NCIO_INT64: return H5T_NATIVE_INT64;
    case NCIO_UINT64: return H5T_NATIVE_UINT64;
    case NCIO_FLOAT: return H5T_NATIVE_FLOAT;
    case NCIO_DOUBLE: return H5T_NATIVE_DOUBLE;
    default: return -1;
  }
}
static hid_t h5_filetype(int t) { /* netCDF-4 writes little-endian by default */
  switch (t) {
    case NCIO_BYTE: return H5T_STD_I8LE;
    case NCIO_UBYTE: return H5T_STD_U8LE;
    case NCIO_SHORT: return H5T_STD_I16LE;
    case NCIO_USHORT: return H5T_STD_U16LE;
    case NCIO_INT: return H5T_STD_I32LE;
    case NCIO_UINT: return H5T_STD_U32LE;
    case NCIO_INT64: return H5T_STD_I64LE;
    case NCIO_UINT64: return H5T_STD_U64LE;
    case NCIO_FLOAT: return H5T_IEEE_F32LE;
    case NCIO_DOUBLE: return H5T_IEEE_F64LE;
    default: return -1;
  }
}
static hid_t h5_chartype(size_t n) { /* caller closes */
  hid_t t = H5Tcopy(H5T_C_S1);
  H5Tset_size(t, n ? n : 1);
  H5Tset_strpad(t, H5T_STR_NULLTERM);
  return t;
}
/* NCIO type of an HDF5 datatype; 0 = not representable (compound, vlen, strings longer than one byte ...) */
static int h5_nctype(hid_t type) {
  const H5T_class_t c = H5Tget_class(type);
  const size_t sz = H5Tget_size(type);
  if (c == H5T_INTEGER) {
    const int u = H5Tget_sign(type) == H5T_SGN_NONE;
    if (sz == 1) return u ? NCIO_UBYTE : NCIO_BYTE;
    if (sz == 2) return u ? NCIO_USHORT : NCIO_SHORT;
    if (sz == 4) return u ? NCIO_UINT : NCIO_INT;
    if (sz == 8) return u ? NCIO_UINT64 : NCIO_INT64;
    return 0;
  }
  if (c == H5T_FLOAT) return sz == 4 ? NCIO_FLOAT : sz == 8 ? NCIO_DOUBLE : 0;
  if (c == H5T_STRING && H5Tis_variable_str(type) <= 0 && sz == 1) return NCIO_CHAR;
  return 0;
}

static void h5_free(ncio_file *f) {
  h5_t *h = (h5_t *)f->h5;
  if (!h) return;
  for (int v = 0; h->dset && v < f->nvars; ++v)
    if (h->dset[v] >= 0) H5Dclose(h->dset[v]);
  for (int d = 0; h->scale && d < h->nscale; ++d)
    if (h->scale_own[d] && h->scale[d] >= 0) H5Dclose(h->scale[d]);
  if (h->file >= 0) H5Fclose(h->file);
  free(h->dset);
  free(h->scale);
  free(h->scale_own);
  free(h);
  f->h5 = NULL;
}

/* ---- reading ---------------------------------------------------------------------------------------------------------- */
static int h5_reserved_att(const char *n) {
  static const char *r[] = {"DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates", "_NCProperties",
                            "_nc3_strict", NULL};
  for (int i = 0; r[i]; ++i)
    if (!strcmp(n, r[i])) return 1;
  return 0;
}
typedef struct { int n; att_t *a; int bad; } h5_attlist;
static herr_t h5_att_cb(hid_t obj, const char *name, const H5A_info_t *info, void *data) {
  (void)info;
  h5_attlist *L = (h5_attlist *)data;
  if (h5_reserved_att(name)) return 0;
  hid_t a = H5Aopen(obj, name, H5P_DEFAULT);
  if (a < 0) return 0;
  hid_t t = H5Aget_type(a), s = H5Aget_space(a);
  const hssize_t np = H5Sget_simple_extent_npoints(s);
  att_t out;
  memset(&out, 0, sizeof(out));
  int ok = 0;
  if (H5Tget_class(t) == H5T_STRING) {
    out.type = NCIO_CHAR;
    if (H5Tis_variable_str(t) > 0) {
      if (np >= 1) {
        char **p = (char **)calloc((size_t)np, sizeof(char *));
        hid_t mt = H5Tget_native_type(t, H5T_DIR_DEFAULT); /* the attribute's own string type (character set included: the library converts none) */
        if (p && H5Aread(a, mt, p) >= 0) {
          const char *str = p[0] ? p[0] : "";
          out.n = (int64_t)strlen(str);
          out.data = calloc((size_t)out.n + 4, 1);
          if (out.data) { memcpy(out.data, str, (size_t)out.n); ok = 1; }
          H5Dvlen_reclaim(mt, s, H5P_DEFAULT, p);
        }
        H5Tclose(mt);
        free(p);
      } else {
        out.data = calloc(4, 1);
        ok = out.data != NULL;
      }
    } else {
      const size_t sz = H5Tget_size(t), tot = sz * (size_t)(np > 0 ? np : 0);
      char *buf = (char *)calloc(tot + 4, 1);
      if (buf && (tot == 0 || H5Aread(a, t, buf) >= 0)) {
        out.n = (int64_t)strnlen(buf, tot);
        out.data = buf;
        ok = 1;
      } else {
        free(buf);
      }
    }
  } else {
    const int nct = h5_nctype(t);
    if (nct && nct != NCIO_CHAR && np >= 0) {
      out.type = nct;
      out.n = (int64_t)np;
      out.data = calloc((size_t)np * 8 + 8, 1);
      if (out.data && (np == 0 || H5Aread(a, h5_memtype(nct), out.data) >= 0)) ok = 1;
      else { free(out.data); out.data = NULL; }
    }
  }
  H5Sclose(s);
  H5Tclose(t);
  H5Aclose(a);
  if (!ok) return 0; /* an attribute of a type the classic model has no word for is left out */
  att_t *na = (att_t *)realloc(L->a, sizeof(att_t) * (size_t)(L->n + 1));
  if (!na) { free(out.data); L->bad = 1; return -1; }
  L->a = na;
  out.name = strdup(name);
  if (!out.name) { free(out.data); L->bad = 1; return -1; }
  L->a[L->n++] = out;
  return 0;
}
static int h5_read_atts(hid_t obj, int *natts, att_t **atts) {
  h5_attlist L = {0, NULL, 0};
  hsize_t idx = 0;
  /* in creation order where the object tracks it (libnetcdf's files and this writer's do: the definition order), by name otherwise */
  if (H5Aiterate2(obj, H5_INDEX_CRT_ORDER, H5_ITER_INC, &idx, h5_att_cb, &L) < 0 && !L.bad) {
    free_atts(L.n, L.a);
    L.n = 0;
    L.a = NULL;
    idx = 0;
    H5Aiterate2(obj, H5_INDEX_NAME, H5_ITER_INC, &idx, h5_att_cb, &L);
  }
  *natts = L.n;
  *atts = L.a;
  return L.bad ? -1 : 0;
}

typedef struct { char **name; hid_t *id; int n, cap; } h5_names;
static herr_t h5_link_cb(hid_t g, const char *name, const H5L_info_t *info, void *data) {
  (void)info;
  h5_names *N = (h5_names *)data;
  hid_t d = H5Dopen2(g, name, H5P_DEFAULT); /* succeeds for datasets only: groups and named types are not part of the classic model */
  if (d < 0) return 0;
  if (N->n == N->cap) {
    const int cap = N->cap ? 2 * N->cap : 64;
    char **nn = (char **)realloc(N->name, sizeof(char *) * (size_t)cap);
    if (nn) N->name = nn;
    hid_t *ni = nn ? (hid_t *)realloc(N->id, sizeof(hid_t) * (size_t)cap) : NULL;
    if (ni) N->id = ni;
    if (!nn || !ni) {   /* out of memory: this dataset is left out (the caller sees a file without it rather than a crash) */
      H5Dclose(d);
      return 0;
    }
    N->cap = cap;
  }
  N->name[N->n] = strdup(name);
  N->id[N->n++] = d;
  return 0;
}
typedef struct { char path[1024]; int found; } h5_scale_hit;
static herr_t h5_scale_cb(hid_t dset, unsigned dim, hid_t scale, void *data) {
  (void)dset; (void)dim;
  h5_scale_hit *hit = (h5_scale_hit *)data;
  if (!hit->found && H5Iget_name(scale, hit->path, sizeof(hit->path)) > 0) hit->found = 1;
  return 1; /* the first attached scale is the dimension */
}
static int h5_find_dim(ncio_file *f, const char *name) {
  for (int d = 0; d < f->ndims; ++d)
    if (!strcmp(f->dims[d].name, name)) return d;
  return -1;
}
static int h5_add_dim(ncio_file *f, const char *name, int64_t len) {
  dim_t *nd = (dim_t *)realloc(f->dims, sizeof(dim_t) * (size_t)(f->ndims + 1));
  if (!nd) return -1;
  f->dims = nd;
  f->dims[f->ndims].name = strdup(name);
  if (!f->dims[f->ndims].name) return -1;
  f->dims[f->ndims].len = len;
  return f->ndims++;
}

static int nc4_open_(const char *path, ncio_file **out) {
  ncio_file *f = (ncio_file *)calloc(1, sizeof(*f));
  h5_t *h = (h5_t *)calloc(1, sizeof(*h));
  if (!f || !h) { free(f); free(h); return fail(NCIO_ENOMEM, "out of memory"); }
  f->h5 = h;
  f->format = 4;
  f->recdim = -1;
  h->file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (h->file < 0) { h5_free(f); free(f); return fail(NCIO_EFORMAT, "ncio_open: libhdf5 cannot open %s (damaged, or not an HDF5 file after all)", path); }
  /* the root group's datasets, in creation order where the file tracks it (libnetcdf's files do), by name otherwise */
  h5_names N = {NULL, NULL, 0, 0};
  hsize_t idx = 0;
  if (H5Literate(h->file, H5_INDEX_CRT_ORDER, H5_ITER_INC, &idx, h5_link_cb, &N) < 0) {
    for (int i = 0; i < N.n; ++i) { free(N.name[i]); H5Dclose(N.id[i]); }
    N.n = 0;
    idx = 0;
    H5Literate(h->file, H5_INDEX_NAME, H5_ITER_INC, &idx, h5_link_cb, &N);
  }
  int rc = 0;
  /* pass 1: dimensions = the dimension scales, numbered by _Netcdf4Dimid where every scale carries one */
  int *is_scale = (int *)calloc((size_t)N.n + 1, sizeof(int)), *dim_only = (int *)calloc((size_t)N.n + 1, sizeof(int));
  int *dimid_att = (int *)calloc((size_t)N.n + 1, sizeof(int));
  int *order = (int *)calloc((size_t)N.n + 1, sizeof(int)); /* scale datasets in dimension order */
  if (!is_scale || !dim_only || !dimid_att || !order) rc = fail(NCIO_ENOMEM, "out of memory");
  for (int i = 0; i < N.n && !rc; ++i)
    if (!N.name[i]) rc = fail(NCIO_ENOMEM, "out of memory");   /* a strdup of the link walk came back empty */
  int nsc = 0, all_numbered = 1;
  for (int i = 0; i < N.n && !rc; ++i) {
    if (H5DSis_scale(N.id[i]) <= 0) continue;
    is_scale[i] = 1;
    ++nsc;
    char nm[256] = "";
    H5DSget_scale_name(N.id[i], nm, sizeof(nm));
    dim_only[i] = !strncmp(nm, NC4_DIM_WITHOUT_VAR, strlen(NC4_DIM_WITHOUT_VAR));
    dimid_att[i] = -1;
    hid_t a = H5Aopen(N.id[i], "_Netcdf4Dimid", H5P_DEFAULT);
    if (a >= 0) {
      int v = -1;
      if (H5Aread(a, H5T_NATIVE_INT, &v) >= 0) dimid_att[i] = v;
      H5Aclose(a);
    }
    if (dimid_att[i] < 0 || dimid_att[i] >= N.n) all_numbered = 0;
  }
  int no = 0;
  if (!rc && all_numbered && nsc > 0) {
    for (int want = 0; want < N.n && no < nsc; ++want)
      for (int i = 0; i < N.n; ++i)
        if (is_scale[i] && dimid_att[i] == want) order[no++] = i;
    if (no != nsc) { no = 0; all_numbered = 0; }
  }
  if (!rc && !all_numbered)
    for (int i = 0; i < N.n; ++i)
      if (is_scale[i]) order[no++] = i;
  for (int k = 0; k < no && !rc; ++k) {
    const int i = order[k];
    hid_t s = H5Dget_space(N.id[i]);
    hsize_t cur[H5S_MAX_RANK] = {0}, mx[H5S_MAX_RANK] = {0};
    const int nd = H5Sget_simple_extent_dims(s, cur, mx);
    H5Sclose(s);
    int64_t len = nd >= 1 ? (int64_t)cur[0] : 1;
    const int unlimited = nd >= 1 && mx[0] == H5S_UNLIMITED;
    const int d = h5_add_dim(f, N.name[i], len);
    if (d < 0) { rc = fail(NCIO_ENOMEM, "out of memory"); break; }
    if (unlimited && f->recdim < 0) { f->recdim = d; f->numrecs = len; }
  }
  /* pass 2: variables = every dataset that is not a dimension without a variable */
  int nvar = 0;
  for (int i = 0; i < N.n && !rc; ++i) nvar += !(is_scale[i] && dim_only[i]);
  f->vars = (var_t *)calloc((size_t)nvar + 1, sizeof(var_t));
  h->dset = (hid_t *)calloc((size_t)nvar + 1, sizeof(hid_t));
  if (!rc && (!f->vars || !h->dset)) rc = fail(NCIO_ENOMEM, "out of memory");
  int phony = 0;
  for (int i = 0; i < N.n && !rc; ++i) {
    if (is_scale[i] && dim_only[i]) continue;
    hid_t t = H5Dget_type(N.id[i]);
    const int nct = h5_nctype(t);
    H5Tclose(t);
    hid_t s = H5Dget_space(N.id[i]);
    hsize_t cur[H5S_MAX_RANK] = {0}, mx[H5S_MAX_RANK] = {0};
    const int nd = H5Sget_simple_extent_type(s) == H5S_SIMPLE ? H5Sget_simple_extent_dims(s, cur, mx) : 0;
    H5Sclose(s);
    if (!nct || nd > NCIO_MAX_DIMS) continue; /* a type or rank the classic model has no word for: not offered */
    var_t *x = &f->vars[f->nvars];
    x->name = strdup(N.name[i]);
    if (!x->name) { rc = fail(NCIO_ENOMEM, "out of memory"); break; }
    x->type = nct;
    x->ndims = nd;
    for (int d = 0; d < nd; ++d) {
      int id = -1;
      if (is_scale[i] && nd == 1) id = h5_find_dim(f, N.name[i]); /* a coordinate variable is its own scale */
      if (id < 0) {
        h5_scale_hit hit;
        hit.found = 0;
        int sidx = 0;
        H5DSiterate_scales(N.id[i], (unsigned)d, &sidx, h5_scale_cb, &hit);
        if (hit.found) {
          const char *base = strrchr(hit.path, '/');
          id = h5_find_dim(f, base ? base + 1 : hit.path);
        }
      }
      if (id < 0) { /* no scale attached (a plain HDF5 file): an anonymous dimension per distinct length, as libnetcdf does */
        for (int q = 0; q < f->ndims && id < 0; ++q)
          if (!strncmp(f->dims[q].name, "phony_dim_", 10) && f->dims[q].len == (int64_t)cur[d] && !(mx[d] == H5S_UNLIMITED)) id = q;
        if (id < 0) {
          char nm[32];
          snprintf(nm, sizeof(nm), "phony_dim_%d", phony++);
          id = h5_add_dim(f, nm, (int64_t)cur[d]);
          if (id < 0) { rc = fail(NCIO_ENOMEM, "out of memory"); break; }
          if (mx[d] == H5S_UNLIMITED && d == 0 && f->recdim < 0) { f->recdim = id; f->numrecs = (int64_t)cur[d]; }
        }
      }
      x->dimids[d] = id;
    }
    if (rc) break;
    if (h5_read_atts(N.id[i], &x->natts, &x->atts)) rc = fail(NCIO_ENOMEM, "out of memory");
    h->dset[f->nvars] = N.id[i];
    N.id[i] = -1;
    f->nvars++;
  }
  if (!rc && h5_read_atts(h->file, &f->ngatts, &f->gatts)) rc = fail(NCIO_ENOMEM, "out of memory");
  for (int i = 0; i < N.n; ++i) {
    free(N.name[i]);
    if (N.id[i] >= 0) H5Dclose(N.id[i]);
  }
  free(N.name); free(N.id); free(is_scale); free(dim_only); free(dimid_att); free(order);
  if (rc) { h5_free(f); free_file(f); return rc; }
  /* the record dimension reports its length through numrecs, as in the classic files (ncio_inq_dim) */
  if (f->recdim >= 0) f->dims[f->recdim].len = 0;
  for (int v = 0; v < f->nvars; ++v) { /* a variable whose unlimited axis is not the first cannot be read record by record: fixed at its length */
    var_t *x = &f->vars[v];
    for (int d = 1; d < x->ndims; ++d)
      if (x->dimids[d] == f->recdim && f->recdim >= 0) {
        h5_free(f);
        free_file(f);
        return fail(NCIO_EFORMAT, "ncio_open: %s: variable with the unlimited dimension not first is not supported", path);
      }
  }
  finish_layout_info(f);
  *out = f;
  return 0;
}

static int nc4_xfer_(ncio_file *f, int varid, int64_t rec, int mem_type, void *buf, int writing, const char *who) {
  h5_t *h = (h5_t *)f->h5;
  var_t *x = &f->vars[varid];
  hid_t d = h->dset[varid];
  hid_t mt, own = -1;
  if (x->type == NCIO_CHAR || mem_type == NCIO_CHAR) {
    if (x->type != mem_type) return fail(NCIO_EINVAL, "%s: %s: text converts to text only", who, x->name);
    mt = own = H5Dget_type(d); /* the bytes as they are: a conversion between 1-byte string types of different padding would keep no character */
  } else {
    mt = h5_memtype(mem_type);
    if (mt < 0) return fail(NCIO_EINVAL, "%s: unsupported memory type %d", who, mem_type);
  }
  /* A value conversion (the host-array flow hands float64 to NF90_FLOAT variables) is done HERE, in one pass over a buffer of the
   * variable's own type, and the library moves that buffer as it is: libhdf5's own conversion works through a 1 MB scratch buffer and
   * is several times slower than the threaded pass here. */
  void *staged = NULL;
  void *user = buf;
  if (x->type != NCIO_CHAR && mem_type != x->type && x->count > 0) {
    staged = malloc((size_t)x->count * (size_t)tsize(x->type));
    if (!staged) return fail(NCIO_ENOMEM, "%s: out of memory", who);
    if (writing && convert_mt(mem_type, user, x->type, staged, x->count)) {
      free(staged);
      return fail(NCIO_EINVAL, "%s: unsupported conversion", who);
    }
    buf = staged;
    mt = h5_memtype(x->type);
  }
  hid_t fs = H5S_ALL, ms = H5S_ALL;
  herr_t e = 0;
  if (x->is_rec) {
    hsize_t cur[H5S_MAX_RANK] = {0}, start[H5S_MAX_RANK] = {0}, cnt[H5S_MAX_RANK] = {0};
    fs = H5Dget_space(d);
    H5Sget_simple_extent_dims(fs, cur, NULL);
    if (rec < 0 || (!writing && rec >= (int64_t)cur[0])) {
      H5Sclose(fs);
      if (own >= 0) H5Tclose(own);
      free(staged);
      return fail(NCIO_ERANGE, "%s: record %lld of %s out of range (%lld records)", who, (long long)rec, x->name, (long long)cur[0]);
    }
    if (writing && rec >= (int64_t)cur[0]) {
      cur[0] = (hsize_t)rec + 1;
      H5Sclose(fs);
      if (H5Dset_extent(d, cur) < 0) { if (own >= 0) H5Tclose(own); free(staged); return fail(NCIO_EIO, "%s: cannot extend %s", who, x->name); }
      fs = H5Dget_space(d);
    }
    start[0] = (hsize_t)rec;
    cnt[0] = 1;
    for (int k = 1; k < x->ndims; ++k) cnt[k] = cur[k];
    e = H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, NULL, cnt, NULL);
    /* the memory side has the record's own shape: with a memory space of another rank (a flat list of the record's elements) the library
     * maps chunks to memory element by element -- 85 MB/s on a 420 MB record, against 1.5 GB/s this way */
    ms = H5Screate_simple(x->ndims, cnt, NULL);
  }
  if (e >= 0 && x->count > 0) e = writing ? H5Dwrite(d, mt, ms, fs, H5P_DEFAULT, buf) : H5Dread(d, mt, ms, fs, H5P_DEFAULT, buf);
  if (fs != H5S_ALL) H5Sclose(fs);
  if (ms != H5S_ALL) H5Sclose(ms);
  if (own >= 0) H5Tclose(own);
  if (e >= 0 && staged && !writing && convert_mt(x->type, staged, mem_type, user, x->count)) {
    free(staged);
    return fail(NCIO_EINVAL, "%s: unsupported conversion", who);
  }
  free(staged);
  if (e < 0) return fail(NCIO_EIO, "%s: libhdf5 could not %s %s", who, writing ? "write" : "read", x->name);
  if (writing && x->is_rec && rec + 1 > f->numrecs) f->numrecs = rec + 1;
  return 0;
}

/* ---- writing ---------------------------------------------------------------------------------------------------------- */
static int nc4_create_(const char *path, ncio_file *f) {
  h5_t *h = (h5_t *)calloc(1, sizeof(*h));
  if (!h) return fail(NCIO_ENOMEM, "out of memory");
  hid_t fcpl = H5Pcreate(H5P_FILE_CREATE); /* definition order survives: link and attribute creation order tracked, as libnetcdf sets it */
  H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
  H5Pset_attr_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
  h->file = H5Fcreate(path, H5F_ACC_TRUNC, fcpl, H5P_DEFAULT);
  H5Pclose(fcpl);
  if (h->file < 0) { free(h); return fail(NCIO_EIO, "ncio_create: libhdf5 cannot create %s", path); }
  f->h5 = h;
  return 0;
}
static int h5_put_atts(hid_t obj, int n, att_t *a, const char *owner) {
  for (int i = 0; i < n; ++i) {
    hid_t t, s, at;
    herr_t e;
    if (a[i].type == NCIO_CHAR) {
      t = h5_chartype((size_t)a[i].n);
      s = H5Screate(a[i].n ? H5S_SCALAR : H5S_NULL);
      at = H5Acreate2(obj, a[i].name, t, s, H5P_DEFAULT, H5P_DEFAULT);
      e = at < 0 ? -1 : (a[i].n ? H5Awrite(at, t, a[i].data) : 0);
      H5Tclose(t);
    } else {
      hsize_t n1 = (hsize_t)a[i].n;
      s = H5Screate_simple(1, &n1, NULL);
      at = H5Acreate2(obj, a[i].name, h5_filetype(a[i].type), s, H5P_DEFAULT, H5P_DEFAULT);
      e = at < 0 ? -1 : (a[i].n ? H5Awrite(at, h5_memtype(a[i].type), a[i].data) : 0);
    }
    if (at >= 0) H5Aclose(at);
    H5Sclose(s);
    if (e < 0) return fail(NCIO_EIO, "ncio_enddef: cannot write attribute %s of %s", a[i].name, owner);
  }
  return 0;
}
static int nc4_enddef_(ncio_file *f) {
  h5_t *h = (h5_t *)f->h5;
  for (int v = 0; v < f->nvars; ++v) { /* the shared per-variable counts (ncio_enddef's first loop) */
    var_t *x = &f->vars[v];
    x->is_rec = x->ndims > 0 && x->dimids[0] == f->recdim;
    x->count = 1;
    for (int d = x->is_rec ? 1 : 0; d < x->ndims; ++d) x->count *= f->dims[x->dimids[d]].len;
  }
  h->dset = (hid_t *)calloc((size_t)f->nvars + 1, sizeof(hid_t));
  h->scale = (hid_t *)calloc((size_t)f->ndims + 1, sizeof(hid_t));
  h->scale_own = (int *)calloc((size_t)f->ndims + 1, sizeof(int));
  if (!h->dset || !h->scale || !h->scale_own) return fail(NCIO_ENOMEM, "out of memory");
  h->nscale = f->ndims;
  for (int v = 0; v < f->nvars; ++v) h->dset[v] = -1;
  for (int d = 0; d < f->ndims; ++d) h->scale[d] = -1;
  int rc = h5_put_atts(h->file, f->ngatts, f->gatts, "the file");
  if (rc) return rc;
  /* a variable named like a dimension must BE its coordinate variable (one axis, that dimension): anything else needs libnetcdf's
   * renaming scheme ("_nc4_non_coord_"), which this writer does not speak */
  int *coord = (int *)calloc((size_t)f->ndims + 1, sizeof(int));
  if (!coord) return fail(NCIO_ENOMEM, "out of memory");
  for (int d = 0; d < f->ndims; ++d) coord[d] = -1;
  for (int v = 0; v < f->nvars; ++v) {
    const int d = h5_find_dim(f, f->vars[v].name);
    if (d < 0) continue;
    if (f->vars[v].ndims != 1 || f->vars[v].dimids[0] != d) {
      free(coord);
      return fail(NCIO_EINVAL, "ncio_enddef: variable %s is named like a dimension but is not its coordinate variable: not expressible here in NetCDF-4",
                  f->vars[v].name);
    }
    coord[d] = v;
  }
  /* datasets in definition order: a dimension's own dataset right before the first variable that needs it would also do; libnetcdf
   * writes dimensions first, so does this */
  for (int d = 0; d < f->ndims && !rc; ++d) {
    if (coord[d] >= 0) continue;
    const int unl = d == f->recdim;
    hsize_t cur = unl ? 0 : (hsize_t)f->dims[d].len, mx = unl ? H5S_UNLIMITED : cur, ch = 1024;
    hid_t s = H5Screate_simple(1, &cur, &mx), dcpl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    if (unl) H5Pset_chunk(dcpl, 1, &ch);
    hid_t ds = H5Dcreate2(h->file, f->dims[d].name, H5T_IEEE_F32BE, s, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    H5Pclose(dcpl);
    H5Sclose(s);
    if (ds < 0) { rc = fail(NCIO_EIO, "ncio_enddef: cannot create dimension %s", f->dims[d].name); break; }
    h->scale[d] = ds;
    h->scale_own[d] = 1;
    char nm[128];
    snprintf(nm, sizeof(nm), "%s%10d", NC4_DIM_WITHOUT_VAR, (int)f->dims[d].len);
    if (H5DSset_scale(ds, nm) < 0) rc = fail(NCIO_EIO, "ncio_enddef: cannot make %s a dimension scale", f->dims[d].name);
  }
  for (int v = 0; v < f->nvars && !rc; ++v) {
    var_t *x = &f->vars[v];
    hsize_t cur[NCIO_MAX_DIMS + 1] = {0}, mx[NCIO_MAX_DIMS + 1] = {0}, ch[NCIO_MAX_DIMS + 1] = {0};
    for (int d = 0; d < x->ndims; ++d) {
      const int unl = x->dimids[d] == f->recdim;
      cur[d] = unl ? 0 : (hsize_t)f->dims[x->dimids[d]].len;
      mx[d] = unl ? H5S_UNLIMITED : cur[d];
      ch[d] = unl ? (x->ndims == 1 ? 512 : 1) : (cur[d] ? cur[d] : 1);
    }
    hid_t s = x->ndims ? H5Screate_simple(x->ndims, cur, mx) : H5Screate(H5S_SCALAR);
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    if (x->is_rec) {
      /* One record per chunk while a record is at most NC4_CHUNK_TARGET; a larger one is cut along its slowest axes into pieces of about
       * that size: whole planes of a [level][y][x] field (a reader that takes one level finds it in one chunk of a few MB), runs of whole
       * rows of an MPAS-shaped [cell][level] field.  (Never one row per chunk: three million 220-byte chunks are a B-tree, not a file.) */
      for (int d = 1; d < x->ndims; ++d) {
        int64_t inner = tsize(x->type);
        for (int q = d + 1; q < x->ndims; ++q) inner *= (int64_t)ch[q];
        if (inner * (int64_t)ch[d] <= NC4_CHUNK_TARGET) break;          /* what is left fits: this axis and the faster ones stay whole */
        if (inner >= NC4_CHUNK_TARGET) { ch[d] = 1; continue; }           /* a single index of this axis is already a chunk's worth: cut the next one too */
        ch[d] = (hsize_t)(NC4_CHUNK_TARGET / inner);                       /* >= 1: as many indices of this axis as fit */
        break;
      }
      H5Pset_chunk(dcpl, x->ndims, ch);
    }
    hid_t own = -1, ft = x->type == NCIO_CHAR ? (own = h5_chartype(1)) : h5_filetype(x->type);
    hid_t ds = H5Dcreate2(h->file, x->name, ft, s, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    if (own >= 0) H5Tclose(own);
    H5Pclose(dcpl);
    H5Sclose(s);
    if (ds < 0) { rc = fail(NCIO_EIO, "ncio_enddef: cannot create variable %s", x->name); break; }
    h->dset[v] = ds;
    const int cd = h5_find_dim(f, x->name);
    if (cd >= 0 && coord[cd] == v) {
      h->scale[cd] = ds;
      if (H5DSset_scale(ds, x->name) < 0) rc = fail(NCIO_EIO, "ncio_enddef: cannot make %s a dimension scale", x->name);
    }
    if (!rc) rc = h5_put_atts(ds, x->natts, x->atts, x->name);
  }
  /* every axis of every variable points at its dimension's scale; the scales are numbered */
  for (int v = 0; v < f->nvars && !rc; ++v) {
    var_t *x = &f->vars[v];
    for (int d = 0; d < x->ndims && !rc; ++d) {
      if (h->scale[x->dimids[d]] == h->dset[v]) continue; /* a coordinate variable is not attached to itself */
      if (H5DSattach_scale(h->dset[v], h->scale[x->dimids[d]], (unsigned)d) < 0)
        rc = fail(NCIO_EIO, "ncio_enddef: cannot attach dimension %s to %s", f->dims[x->dimids[d]].name, x->name);
    }
  }
  for (int d = 0; d < f->ndims && !rc; ++d) {
    hid_t s = H5Screate(H5S_SCALAR), a = H5Acreate2(h->scale[d], "_Netcdf4Dimid", H5T_NATIVE_INT, s, H5P_DEFAULT, H5P_DEFAULT);
    int id = d;
    if (a < 0 || H5Awrite(a, H5T_NATIVE_INT, &id) < 0) rc = fail(NCIO_EIO, "ncio_enddef: cannot number dimension %s", f->dims[d].name);
    if (a >= 0) H5Aclose(a);
    H5Sclose(s);
  }
  free(coord);
  if (!rc) f->defmode = 0;
  return rc;
}
static int nc4_close_(ncio_file *f) {
  h5_t *h = (h5_t *)f->h5;
  int rc = 0;
  if (f->writing && !f->defmode && f->recdim >= 0 && h && h->dset) {
    /* every variable along the unlimited dimension, and the dimension itself, end at the same length */
    for (int v = 0; v < f->nvars; ++v) {
      var_t *x = &f->vars[v];
      if (!x->is_rec || h->dset[v] < 0) continue;
      hsize_t cur[H5S_MAX_RANK] = {0};
      hid_t s = H5Dget_space(h->dset[v]);
      H5Sget_simple_extent_dims(s, cur, NULL);
      H5Sclose(s);
      if ((int64_t)cur[0] != f->numrecs) {
        cur[0] = (hsize_t)f->numrecs;
        if (H5Dset_extent(h->dset[v], cur) < 0) rc = fail(NCIO_EIO, "ncio_close: cannot size %s", x->name);
      }
    }
    if (h->scale_own[f->recdim]) {
      hsize_t n = (hsize_t)f->numrecs;
      if (H5Dset_extent(h->scale[f->recdim], &n) < 0) rc = fail(NCIO_EIO, "ncio_close: cannot size the unlimited dimension");
    }
  }
  if (h && h->file >= 0 && f->writing && H5Fflush(h->file, H5F_SCOPE_GLOBAL) < 0) rc = fail(NCIO_EIO, "ncio_close: flush failed");
  h5_free(f);
  return rc;
}
/* libhdf5 is not thread-safe in its usual build, and hosts do use two threads (io_nc.run_series reads the next file while the current one is
 * written): every entry into the library goes through one lock. */
static pthread_mutex_t g_h5_lock = PTHREAD_MUTEX_INITIALIZER;
/* Failures are reported through ncio_strerror, not printed by libhdf5 -- but only while THIS code is inside the library: the host process's
 * own error handler (another HDF5 user's printing, a test harness's hook) is saved on entry and put back on the way out. */
#define NC4_LOCKED(call)                                   \
  do {                                                     \
    pthread_mutex_lock(&g_h5_lock);                        \
    H5E_auto2_t efn_ = NULL;                               \
    void *edata_ = NULL;                                   \
    H5Eget_auto2(H5E_DEFAULT, &efn_, &edata_);             \
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);                 \
    int rc_ = (call);                                      \
    H5Eset_auto2(H5E_DEFAULT, efn_, edata_);               \
    pthread_mutex_unlock(&g_h5_lock);                      \
    return rc_;                                            \
  } while (0)
static int nc4_open(const char *path, ncio_file **out) { NC4_LOCKED(nc4_open_(path, out)); }
static int nc4_xfer(ncio_file *f, int varid, int64_t rec, int mem_type, void *buf, int writing, const char *who) {
  NC4_LOCKED(nc4_xfer_(f, varid, rec, mem_type, buf, writing, who));
}
static int nc4_create(const char *path, ncio_file *f) { NC4_LOCKED(nc4_create_(path, f)); }
static int nc4_enddef(ncio_file *f) { NC4_LOCKED(nc4_enddef_(f)); }
static int nc4_close(ncio_file *f) { NC4_LOCKED(nc4_close_(f)); }
#define NC4_AVAILABLE 1
#else /* ---- built without HDF5: every entry says so ------------------------------------------------------------------------ */
#define NC4_AVAILABLE 0
static int nc4_missing(const char *path) {
  return fail(NCIO_EHDF5, "%s is (or is to be) a NetCDF-4/HDF5 file, and this build of libmpassit_ncio has no HDF5 (rebuild with MPASSIT_HDF5_ROOT "
                          "pointing at hdf5.h / libhdf5 / libhdf5_hl, or convert: `nccopy -k cdf5 in.nc out.nc`)", path);
}
static int nc4_open(const char *path, ncio_file **out) { (void)out; return nc4_missing(path); }
static int nc4_create(const char *path, ncio_file *f) { (void)f; return nc4_missing(path); }
static int nc4_xfer(ncio_file *f, int varid, int64_t rec, int mem_type, void *buf, int writing, const char *who) {
  (void)f; (void)varid; (void)rec; (void)mem_type; (void)buf; (void)writing;
  return nc4_missing(who);
}
static int nc4_enddef(ncio_file *f) { (void)f; return nc4_missing("the output"); }
static int nc4_close(ncio_file *f) { (void)f; return 0; }
static void h5_free(ncio_file *f) { (void)f; }
#endif
