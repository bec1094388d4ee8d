/* rsys_h5.h -- C ABI of the shard / embedding file adapter (librsys_h5.so): blosc-compressed HDF5 datasets, the
 * file format at the two data seams of the training path.
 *
 *   written by   notebooks/Training/transformer.jl:73-77   media_embeddings.h5: `file[k, blosc = 3] = v`
 *                notebooks/Training/transformer.jl:228-231 {training,test}/{shard}/{p}.h5: 27 flat per-token datasets
 *                notebooks/Training/transformer.jl:190-200 pad.h5
 *   read by      notebooks/Training/transformer.py:86-89   PretrainDataset: `for k in f: d[k] = f[k][:]` (h5py + hdf5plugin)
 *                notebooks/Training/transformer.py:131-140 FinetuneDataset
 *                notebooks/Training/transformer.model.py:379-389 load_pretrained_embeddings
 *
 * Host-side only: links the image's libhdf5 (1.10.6) and c-blosc (1.21); no GPU code, no dependency on librsys_hip.so.
 * The library registers HDF5 filter 32001 ("blosc") itself; the same object file is also a loadable HDF5 filter
 * plugin (H5PLget_plugin_type / H5PLget_plugin_info), so any libhdf5 client finds the filter through HDF5_PLUGIN_PATH.
 *
 * Every function returns 0 on success, -1 bad argument, -5 HDF5 / blosc error; rsys_h5_last_error() has the text.
 * Dataset shapes are reported in the file's (row-major, C) dimension order: an array Julia writes as (M, V)
 * column-major appears as (V, M), which is what h5py hands the reference.
 */
#ifndef RSYS_H5_H
#define RSYS_H5_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of the datasets on the path */
enum rsys_h5_dtype {
  RSYS_H5_F32 = 0, RSYS_H5_F64 = 1, RSYS_H5_I32 = 2, RSYS_H5_I64 = 3, RSYS_H5_U8 = 4,
  RSYS_H5_I8 = 5, RSYS_H5_I16 = 6, RSYS_H5_U16 = 7, RSYS_H5_U32 = 8, RSYS_H5_U64 = 9
};
#define RSYS_H5_MAX_DIMS 4

/* h5py.File(fn) / HDF5.h5open(fn, "w"): mode 0 = read only, 1 = create / truncate */
int rsys_h5_open(const char* path, int mode, void** file);
int rsys_h5_close(void* file);

/* `for k in f` (transformer.py:88): datasets of the root group, in HDF5's name order */
int rsys_h5_num_datasets(void* file, int32_t* n);
int rsys_h5_dataset_name(void* file, int32_t index, char* name, int32_t capacity);

/* dtype / rank / dims / blosc level (-1: not blosc-filtered) of one dataset */
int rsys_h5_dataset_info(void* file, const char* name, int32_t* dtype, int32_t* ndim, int64_t* dims, int32_t* blosc_level);

/* `f[k][:]`: the whole dataset into dst (dst_bytes must equal its size) */
int rsys_h5_read(void* file, const char* name, void* dst, int64_t dst_bytes);

/* `file[k, blosc = level] = v` (transformer.jl:75, 198, 230): chunked, byte-shuffled, blosclz at `level`;
 * level < 0 writes a contiguous uncompressed dataset.  Empty datasets and rank-0 (scalar) datasets are always written
 * contiguous; ndim = 0 stores one value (h5py's `create_dataset(k, data=python_float)`, Finetune/register.py:34-36). */
int rsys_h5_write(void* file, const char* name, int32_t dtype, int32_t ndim, const int64_t* dims, const void* src,
                  int32_t blosc_level);

const char* rsys_h5_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
