/* C ABI of include/rsys_h5.h over libhdf5: whole-dataset reads and blosc-filtered writes of the root-group datasets
 * that make up a training shard (transformer.jl:228-231 / transformer.py:86-89) and media_embeddings.h5. */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hdf5.h>

#include "rsys_h5.h"

#define RSYS_BLOSC_FILTER_ID 32001
#define RSYS_H5_BADARG (-1)
#define RSYS_H5_ERROR (-5)
#define CHUNK_TARGET_BYTES ((hsize_t)8 << 20)

int rsys_h5_register_blosc(void);

static __thread char g_err[512];

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

const char* rsys_h5_last_error(void) { return g_err; }

typedef struct { hid_t fid; int writable; } h5file;

static int library_ready(void) {
  static int ready = 0;
  if (ready) return 0;
  if (H5open() < 0) return fail(RSYS_H5_ERROR, "H5open failed");
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);         /* errors are reported through return codes, not stderr */
  if (rsys_h5_register_blosc() != 0) return fail(RSYS_H5_ERROR, "cannot register HDF5 filter 32001 (blosc)");
  ready = 1;
  return 0;
}

int rsys_h5_open(const char* path, int mode, void** file) {
  if (!path || !file || (mode != 0 && mode != 1)) return fail(RSYS_H5_BADARG, "rsys_h5_open: bad argument");
  int rc = library_ready();
  if (rc) return rc;
  hid_t fid = mode == 0 ? H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT) : H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (fid < 0) return fail(RSYS_H5_ERROR, "cannot %s %s", mode == 0 ? "open" : "create", path);
  h5file* f = (h5file*)malloc(sizeof *f);
  f->fid = fid;
  f->writable = mode;
  *file = f;
  return 0;
}

int rsys_h5_close(void* file) {
  if (!file) return 0;
  h5file* f = (h5file*)file;
  herr_t e = H5Fclose(f->fid);
  free(f);
  return e < 0 ? fail(RSYS_H5_ERROR, "H5Fclose failed") : 0;
}

/* index-th dataset of the root group in name order; index < 0 only counts.  Returns the count or a negative code. */
static long long walk_datasets(h5file* f, int32_t index, char* name, int32_t capacity) {
  H5G_info_t gi;
  if (H5Gget_info(f->fid, &gi) < 0) return fail(RSYS_H5_ERROR, "H5Gget_info failed");
  long long seen = 0;
  char buf[256];
  for (hsize_t i = 0; i < gi.nlinks; ++i) {
    if (H5Lget_name_by_idx(f->fid, ".", H5_INDEX_NAME, H5_ITER_INC, i, buf, sizeof buf, H5P_DEFAULT) < 0)
      return fail(RSYS_H5_ERROR, "H5Lget_name_by_idx failed at link %llu", (unsigned long long)i);
    H5O_info_t oi;
    if (H5Oget_info_by_name(f->fid, buf, &oi, H5P_DEFAULT) < 0 || oi.type != H5O_TYPE_DATASET) continue;
    if (seen == index) {
      if ((int32_t)strlen(buf) + 1 > capacity) return fail(RSYS_H5_BADARG, "name buffer too small for %s", buf);
      strcpy(name, buf);
      return seen;
    }
    ++seen;
  }
  if (index >= 0) return fail(RSYS_H5_BADARG, "dataset index %d out of range (%lld datasets)", index, seen);
  return seen;
}

int rsys_h5_num_datasets(void* file, int32_t* n) {
  if (!file || !n) return fail(RSYS_H5_BADARG, "rsys_h5_num_datasets: bad argument");
  long long c = walk_datasets((h5file*)file, -1, NULL, 0);
  if (c < 0) return (int)c;
  *n = (int32_t)c;
  return 0;
}

int rsys_h5_dataset_name(void* file, int32_t index, char* name, int32_t capacity) {
  if (!file || !name || index < 0 || capacity <= 0) return fail(RSYS_H5_BADARG, "rsys_h5_dataset_name: bad argument");
  long long c = walk_datasets((h5file*)file, index, name, capacity);
  return c < 0 ? (int)c : 0;
}

static hid_t native_type(int32_t dtype) {
  switch (dtype) {
    case RSYS_H5_F32: return H5T_NATIVE_FLOAT;
    case RSYS_H5_F64: return H5T_NATIVE_DOUBLE;
    case RSYS_H5_I32: return H5T_NATIVE_INT32;
    case RSYS_H5_I64: return H5T_NATIVE_INT64;
    case RSYS_H5_U8: return H5T_NATIVE_UINT8;
    case RSYS_H5_I8: return H5T_NATIVE_INT8;
    case RSYS_H5_I16: return H5T_NATIVE_INT16;
    case RSYS_H5_U16: return H5T_NATIVE_UINT16;
    case RSYS_H5_U32: return H5T_NATIVE_UINT32;
    case RSYS_H5_U64: return H5T_NATIVE_UINT64;
    default: return -1;
  }
}

static const int dtype_bytes[10] = {4, 8, 4, 8, 1, 1, 2, 2, 4, 8};

static int classify(hid_t type, int32_t* dtype) {
  const H5T_class_t cls = H5Tget_class(type);
  const size_t size = H5Tget_size(type);
  if (cls == H5T_FLOAT) {
    if (size == 4) { *dtype = RSYS_H5_F32; return 0; }
    if (size == 8) { *dtype = RSYS_H5_F64; return 0; }
  } else if (cls == H5T_INTEGER) {
    const int is_signed = H5Tget_sign(type) != H5T_SGN_NONE;
    switch (size) {
      case 1: *dtype = is_signed ? RSYS_H5_I8 : RSYS_H5_U8; return 0;
      case 2: *dtype = is_signed ? RSYS_H5_I16 : RSYS_H5_U16; return 0;
      case 4: *dtype = is_signed ? RSYS_H5_I32 : RSYS_H5_U32; return 0;
      case 8: *dtype = is_signed ? RSYS_H5_I64 : RSYS_H5_U64; return 0;
    }
  }
  return -1;
}

int rsys_h5_dataset_info(void* file, const char* name, int32_t* dtype, int32_t* ndim, int64_t* dims, int32_t* blosc_level) {
  if (!file || !name || !dtype || !ndim || !dims) return fail(RSYS_H5_BADARG, "rsys_h5_dataset_info: bad argument");
  h5file* f = (h5file*)file;
  hid_t ds = H5Dopen2(f->fid, name, H5P_DEFAULT);
  if (ds < 0) return fail(RSYS_H5_ERROR, "no dataset %s", name);
  int rc = 0;
  hid_t type = H5Dget_type(ds), space = H5Dget_space(ds), dcpl = H5Dget_create_plist(ds);
  hsize_t hd[H5S_MAX_RANK];
  const int rank = H5Sget_simple_extent_ndims(space);
  if (classify(type, dtype) != 0) rc = fail(RSYS_H5_ERROR, "dataset %s: element type is not a plain integer or float", name);
  else if (rank < 0 || rank > RSYS_H5_MAX_DIMS) rc = fail(RSYS_H5_ERROR, "dataset %s: rank %d not supported", name, rank);
  else {
    H5Sget_simple_extent_dims(space, hd, NULL);
    *ndim = rank;
    for (int i = 0; i < rank; ++i) dims[i] = (int64_t)hd[i];
    if (blosc_level) {
      *blosc_level = -1;
      const int nf = H5Pget_nfilters(dcpl);
      for (int i = 0; i < nf; ++i) {
        unsigned flags, cfg, values[8] = {0};
        size_t nelem = 8;
        char fname[16];
        if (H5Pget_filter2(dcpl, (unsigned)i, &flags, &nelem, values, sizeof fname, fname, &cfg) == RSYS_BLOSC_FILTER_ID)
          *blosc_level = nelem >= 5 ? (int32_t)values[4] : 5;
      }
    }
  }
  H5Pclose(dcpl); H5Sclose(space); H5Tclose(type); H5Dclose(ds);
  return rc;
}

int rsys_h5_read(void* file, const char* name, void* dst, int64_t dst_bytes) {
  if (!file || !name || dst_bytes < 0 || (!dst && dst_bytes)) return fail(RSYS_H5_BADARG, "rsys_h5_read: bad argument");
  int32_t dtype, ndim;
  int64_t dims[RSYS_H5_MAX_DIMS];
  int rc = rsys_h5_dataset_info(file, name, &dtype, &ndim, dims, NULL);
  if (rc) return rc;
  int64_t count = 1;
  for (int i = 0; i < ndim; ++i) count *= dims[i];
  if (count * dtype_bytes[dtype] != dst_bytes)
    return fail(RSYS_H5_BADARG, "dataset %s holds %lld bytes, destination has %lld", name,
                (long long)(count * dtype_bytes[dtype]), (long long)dst_bytes);
  if (count == 0) return 0;
  h5file* f = (h5file*)file;
  hid_t ds = H5Dopen2(f->fid, name, H5P_DEFAULT);
  if (ds < 0) return fail(RSYS_H5_ERROR, "no dataset %s", name);
  herr_t e = H5Dread(ds, native_type(dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, dst);
  H5Dclose(ds);
  return e < 0 ? fail(RSYS_H5_ERROR, "H5Dread failed on %s (corrupt chunk or missing filter)", name) : 0;
}

int rsys_h5_write(void* file, const char* name, int32_t dtype, int32_t ndim, const int64_t* dims, const void* src,
                  int32_t blosc_level) {
  if (!file || !name || (!dims && ndim > 0) || ndim < 0 || ndim > RSYS_H5_MAX_DIMS || native_type(dtype) < 0 || blosc_level > 9)
    return fail(RSYS_H5_BADARG, "rsys_h5_write: bad argument");
  h5file* f = (h5file*)file;
  if (!f->writable) return fail(RSYS_H5_BADARG, "file was opened read-only");
  hsize_t hd[RSYS_H5_MAX_DIMS], chunk[RSYS_H5_MAX_DIMS];
  hsize_t count = 1, inner = dtype_bytes[dtype];
  for (int i = 0; i < ndim; ++i) {
    if (dims[i] < 0) return fail(RSYS_H5_BADARG, "negative dimension");
    hd[i] = (hsize_t)dims[i];
    count *= hd[i];
  }
  if (count && !src) return fail(RSYS_H5_BADARG, "rsys_h5_write: null source");
  hid_t space = ndim == 0 ? H5Screate(H5S_SCALAR) : H5Screate_simple(ndim, hd, NULL);   /* rank 0: one value, as h5py stores a Python float */
  hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
  int rc = 0;
  if (blosc_level >= 0 && count > 0 && ndim > 0) {
    /* whole trailing dimensions, as many leading rows as fit the chunk target */
    for (int i = 1; i < ndim; ++i) inner *= hd[i];
    for (int i = 0; i < ndim; ++i) chunk[i] = hd[i];
    hsize_t lead = CHUNK_TARGET_BYTES / (inner ? inner : 1);
    if (lead < 1) lead = 1;
    if (chunk[0] > lead) chunk[0] = lead;
    const unsigned values[7] = {0, 0, 0, 0, (unsigned)blosc_level, 1 /* byte shuffle */, 0 /* blosclz */};
    if (H5Pset_chunk(dcpl, ndim, chunk) < 0 ||
        H5Pset_filter(dcpl, RSYS_BLOSC_FILTER_ID, H5Z_FLAG_OPTIONAL, 7, values) < 0)
      rc = fail(RSYS_H5_ERROR, "cannot set up the blosc pipeline for %s", name);
  }
  if (!rc) {
    hid_t ds = H5Dcreate2(f->fid, name, native_type(dtype), space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    if (ds < 0) rc = fail(RSYS_H5_ERROR, "cannot create dataset %s", name);
    else {
      if (count && H5Dwrite(ds, native_type(dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, src) < 0)
        rc = fail(RSYS_H5_ERROR, "H5Dwrite failed on %s", name);
      H5Dclose(ds);
    }
  }
  H5Pclose(dcpl);
  H5Sclose(space);
  return rc;
}
