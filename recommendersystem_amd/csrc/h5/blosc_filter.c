/* HDF5 filter 32001 ("blosc") over c-blosc 1.x: one chunk = one blosc buffer.
 *
 * This is the filter behind `file[k, blosc = 3] = v` in notebooks/Training/transformer.jl:75,198,230 (HDF5.jl's
 * H5Zblosc) and behind `import hdf5plugin` in notebooks/Training/transformer.py:11; neither package is vendored in
 * the reference, so the filter is restated here from its registered public convention:
 *
 *   cd_values[0] filter revision (2)      cd_values[4] compression level 0..9   (user)
 *   cd_values[1] blosc format version (2) cd_values[5] shuffle 0 / 1 byte / 2 bit (user)
 *   cd_values[2] element size in bytes    cd_values[6] compressor code, 0 = blosclz (user)
 *   cd_values[3] chunk size in bytes
 *
 * [0..3] are filled in by the set_local callback when the dataset is created.  A chunk blosc cannot shrink is left
 * to HDF5 (the filter is registered optional, HDF5 then stores the chunk raw and flags it in the chunk's filter mask).
 * Built twice: into librsys_h5.so (registered with H5Zregister) and as h5plugin/libH5Zblosc.so, a loadable filter
 * plugin for any other libhdf5 client (HDF5_PLUGIN_PATH). */
#include <stdlib.h>
#include <string.h>

#include <blosc.h>
#include <hdf5.h>
#include <H5PLextern.h>

#define RSYS_BLOSC_FILTER_ID 32001
#define RSYS_BLOSC_FILTER_REVISION 2

static herr_t rsys_blosc_set_local(hid_t dcpl, hid_t type, hid_t space) {
  (void)space;
  unsigned flags = 0;
  size_t nelem = 8;
  unsigned values[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  char name[8];
  if (H5Pget_filter_by_id2(dcpl, RSYS_BLOSC_FILTER_ID, &flags, &nelem, values, sizeof name, name, NULL) < 0) return -1;
  if (nelem < 4) nelem = 4;                       /* user values [4..] stay as given; absent ones read as defaults */
  values[0] = RSYS_BLOSC_FILTER_REVISION;
  values[1] = BLOSC_VERSION_FORMAT;

  hsize_t chunk[32];
  const int ndim = H5Pget_chunk(dcpl, 32, chunk);
  if (ndim < 0 || ndim > 32) return -1;

  size_t typesize = H5Tget_size(type);
  if (typesize == 0) return -1;
  size_t basesize = typesize;
  if (H5Tget_class(type) == H5T_ARRAY) {         /* shuffle on the element of an array type */
    hid_t super = H5Tget_super(type);
    basesize = H5Tget_size(super);
    H5Tclose(super);
  }
  if (basesize > BLOSC_MAX_TYPESIZE) basesize = 1;
  values[2] = (unsigned)basesize;

  size_t bytes = typesize;
  for (int i = 0; i < ndim; ++i) bytes *= (size_t)chunk[i];
  values[3] = (unsigned)bytes;
  return H5Pmodify_filter(dcpl, RSYS_BLOSC_FILTER_ID, flags, nelem, values);
}

static size_t rsys_blosc_filter(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes,
                                size_t* buf_size, void** buf) {
  void* out = NULL;
  size_t out_size = 0;
  int status = 0;

  if (flags & H5Z_FLAG_REVERSE) {
    size_t cbytes = 0, blocksize = 0;
    blosc_cbuffer_sizes(*buf, &out_size, &cbytes, &blocksize);   /* sizes from the 16-byte blosc header */
    if (cbytes > nbytes || out_size == 0) return 0;
    out = malloc(out_size);
    if (!out) return 0;
    status = blosc_decompress_ctx(*buf, out, out_size, 1);
    if (status <= 0 || (size_t)status != out_size) { free(out); return 0; }
  } else {
    if (cd_nelmts < 4) return 0;
    const size_t typesize = cd_values[2];
    const int clevel = cd_nelmts >= 5 ? (int)cd_values[4] : 5;
    const int shuffle = cd_nelmts >= 6 ? (int)cd_values[5] : 1;
    const char* compressor = BLOSC_BLOSCLZ_COMPNAME;
    if (cd_nelmts >= 7 && blosc_compcode_to_compname((int)cd_values[6], &compressor) < 0) return 0;
    out_size = nbytes;                                           /* must shrink: no BLOSC_MAX_OVERHEAD allowance */
    out = malloc(out_size);
    if (!out) return 0;
    status = blosc_compress_ctx(clevel, shuffle, typesize, nbytes, *buf, out, out_size, compressor, 0, 1);
    if (status <= 0) { free(out); return 0; }                    /* 0: incompressible, HDF5 keeps the raw chunk */
  }
  free(*buf);
  *buf = out;
  *buf_size = out_size;
  return (size_t)status;
}

const H5Z_class2_t rsys_blosc_class = {
    H5Z_CLASS_T_VERS, (H5Z_filter_t)RSYS_BLOSC_FILTER_ID, 1, 1, "blosc", NULL, rsys_blosc_set_local, rsys_blosc_filter};

/* registers the filter with the libhdf5 this object is linked against (idempotent) */
int rsys_h5_register_blosc(void) {
  if (H5Zfilter_avail(RSYS_BLOSC_FILTER_ID) > 0) return 0;
  return H5Zregister(&rsys_blosc_class) < 0 ? -1 : 0;
}

H5PL_type_t H5PLget_plugin_type(void) { return H5PL_TYPE_FILTER; }
const void* H5PLget_plugin_info(void) { return &rsys_blosc_class; }
