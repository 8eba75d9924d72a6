// Input path of the train loop (utils/dataset_utils.py:138-283 of the reference: TFRecordDataset -> parse_single_sequence_example
// -> (x - mean) / std -> padded batch), host side in C (csrc/host.cpp, plain C++: it also builds under the sanitizers) and one HIP
// kernel (this file):
//   las_tfrecord_index        TFRecord framing of a file image (length, masked crc32c, payload, masked crc32c) + the frame /
//                             label counts of every SequenceExample, one pass, optional CRC check (hardware crc32c)
//   las_tfrecord_parse_batch  SequenceExample{feature_lists{'inputs': float_list per frame, 'labels': bytes_list per step}}
//                             -> packed [sum T, F] float frames + the label tokens as bytes with offsets
//   las_normalize_pad_bf16    packed frames -> (x - mean) / std -> bf16 [B, T', F'] zero padded (the listener's input layout)
// The Python host (utils/fast_input.py) keeps the reference's pipeline semantics (repeat, shuffle buffer, filters, padded
// batches) on record INDICES and calls these once per batch from a prefetch thread.
#include "las_common.h"

namespace {

// (x - mean) / std in double (what numpy does with the float64 norm.dmp arrays), rounded to float, then to bf16
__global__ __launch_bounds__(256) void normalize_pad_kernel(const float* __restrict__ frames, const int64_t* __restrict__ row_off,
                                                            const double* __restrict__ mean, const double* __restrict__ stdv,
                                                            int F, unsigned short* __restrict__ out, int B, int Tp, int Fp,
                                                            int32_t* __restrict__ lengths) {
  const int b = blockIdx.y;
  const int64_t r0 = row_off[b];
  const int T = (int)(row_off[b + 1] - r0);
  if (lengths && blockIdx.x == 0 && threadIdx.x == 0) lengths[b] = min(T, Tp);
  const int64_t total = (int64_t)Tp * Fp;
  unsigned short* o = out + (int64_t)b * total;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i / Fp), f = (int)(i - (int64_t)t * Fp);
    float v = 0.f;
    if (t < T && f < F) {
      const float x = frames[(r0 + t) * F + f];
      v = mean ? (float)(((double)x - mean[f]) / stdv[f]) : x;
    }
    o[i] = las_f2bf(v);
  }
}

}  // namespace

extern "C" int las_normalize_pad_bf16(const float* frames, const int64_t* frame_row_offsets, const double* mean, const double* stdv,
                                      int num_channels, las_bf16* out, int B, int T_padded, int F_padded, int32_t* lengths_out,
                                      void* stream) {
  LAS_REQUIRE(frames && frame_row_offsets && out && B > 0 && T_padded > 0 && F_padded >= num_channels && num_channels > 0 &&
                  ((mean == nullptr) == (stdv == nullptr)),
              "las_normalize_pad_bf16: bad arguments");
  const int64_t total = (int64_t)T_padded * F_padded;
  int bx = (int)((total + 255) / 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(normalize_pad_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, frames, frame_row_offsets, mean, stdv,
                     num_channels, out, B, T_padded, F_padded, lengths_out);
  LAS_LAUNCH_CHECK("normalize + pad launch");
  return LAS_OK;
}
