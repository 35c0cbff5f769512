/*
 * transoar_optim.h -- C ABI of the multi-tensor AdamW update of the training step
 * (reference: torch.optim.AdamW as set up in scripts/train.py:52-63 -- two parameter groups, decoupled weight decay,
 * no amsgrad).  One launch updates every fp32 parameter of the model: 28 bytes of HBM traffic per parameter
 * (p, g, m, v in; p, m, v out).  Device pointers, asynchronous on `hip_stream`, capturable (learning rates and the
 * step count are read from device memory); returns 0, a hipError_t, -1 (NULL) or -2 (bad size).
 */
#ifndef TRANSOAR_OPTIM_H
#define TRANSOAR_OPTIM_H
#ifdef __cplusplus
extern "C" {
#endif

/* One tensor of the update.  All pointers are device pointers, 16-byte aligned; lr and step point to single floats
 * (the group's learning rate; the number of this update, >= 1: the caller has already incremented it). */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  const float* lr;
  const float* step;
  long n;           /* elements */
  long reserved;
} transoar_adamw_tensor;

/* Work list of the launch: chunk i covers elements [chunk_offset[i], chunk_offset[i] + TRANSOAR_ADAMW_CHUNK) of tensor
 * chunk_tensor[i].  `tensors`, `chunk_tensor`, `chunk_offset` are device arrays built once by the caller. */
#define TRANSOAR_ADAMW_CHUNK 16384

/* p -= lr*wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps) */
int transoar_adamw_step(const transoar_adamw_tensor* tensors, const int* chunk_tensor, const long* chunk_offset, int n_chunks,
                        double beta1, double beta2, float eps, float weight_decay, void* hip_stream);

int transoar_optim_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
