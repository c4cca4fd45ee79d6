/*
 * rt_mi355x_rccl.h -- C ABI of the multi-GPU leg of the MI355X sphere path tracer
 * (librt_mi355x_rccl.so: RCCL over xGMI, one process per GPU).
 *
 * The reference renders on one device (src/vk/VulkanContext.cpp:413-414), so
 * nothing here replaces a reference interface; these calls are what a host in
 * the reference's language (C++, src/core/App.cpp:403) adds around
 * rt_render_stripes / rt_render_shard (rt_mi355x.h) when the image is shared out
 * over the GPUs of a node.  Pixels are independent (raytrace.comp.glsl:333-335
 * touches the invocation's own texel only) and a texel's random stream depends
 * on (x, y, frame) alone (:310-311), so there are two exact ways to split a step:
 *
 *   stripes  rank r of G renders the stripes r, r + G, ... of stripe_rows rows of
 *            every frame; the accumulators are put together by ONE all-gather of
 *            equal-sized packed shards.  Bit-identical to one GPU.
 *   frames   rank r renders the frames r, r + G, ... of the whole image into its
 *            own accumulator; the accumulators are summed (all-reduce, fp32) and
 *            alpha is set back to 1 (the shader stores alpha = 1 with every
 *            frame, :335; the sum holds G).  Streams identical to one GPU, the
 *            cross-frame sum associated differently (a few ulp).
 *
 * Packing, de-interleaving and the alpha fix are HIP kernels of this library;
 * the collectives are ncclAllGather / ncclAllReduce on the communicator the
 * caller brings (or creates through the two calls below).  Every call is
 * asynchronous on the stream it is given; all functions return 0 or a non-zero
 * rt_status (rt_mi355x.h), nothing throws.  A communicator object is used from
 * one thread at a time.
 */
#ifndef RT_MI355X_RCCL_H
#define RT_MI355X_RCCL_H

#include <stddef.h>
#include <stdint.h>

#include "rt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_comm *rt_comm_handle;

#define RT_COMM_ID_BYTES 128

/* ncclGetUniqueId: rank 0 calls this and hands the 128 bytes to every other
 * rank (over whatever the host program has: a socket, MPI, torch.distributed). */
int rt_comm_unique_id(unsigned char id[RT_COMM_ID_BYTES]);

/* ncclCommInitRank on HIP device device_id, plus the object that keeps this
 * library's staging buffers.  Collective: every rank of the job calls it. */
int rt_comm_create(const unsigned char id[RT_COMM_ID_BYTES], int world, int rank, int device_id, rt_comm_handle *out);

/* The same around a communicator the caller already has (an ncclComm_t of
 * `world` ranks in which this process is `rank`); the caller keeps ownership. */
int rt_comm_adopt(void *nccl_comm, int world, int rank, int device_id, rt_comm_handle *out);

/* Frees the staging buffers and, for rt_comm_create, the communicator. */
int rt_comm_destroy(rt_comm_handle c);

int rt_comm_world(rt_comm_handle c);
int rt_comm_rank(rt_comm_handle c);

/* Stripe split in one call: packs this rank's stripes of h's accumulator,
 * all-gathers the ranks' packed stripes and writes the complete image to
 * dst_image (width * height float4 on this GPU; may be h's accumulator itself),
 * all on `stream`.  stripe_rows as passed to rt_render_stripes by every rank
 * (with stripe_first = rank, stripe_stride = world). */
int rt_gather_stripes(rt_handle h, rt_comm_handle c, uint32_t stripe_rows, void *dst_image, void *stream);

/* Frame split in one call: copies h's accumulator to dst_image (may be the
 * accumulator itself), sums the ranks' images there (fp32 all-reduce) and sets
 * alpha to 1, on `stream`. */
int rt_reduce_frames(rt_handle h, rt_comm_handle c, void *dst_image, void *stream);

/* The two halves of each, for callers that pipeline steps: the first half reads
 * the accumulator into staging buffer `slot` (0 or 1) on the stream that
 * renders, after which the next step may render; the second half runs the
 * collective from that buffer on another stream (the caller orders the two
 * streams with an event, and does not reuse a slot before its collective has
 * finished). */
int rt_comm_pack_stripes(rt_handle h, rt_comm_handle c, uint32_t stripe_rows, uint32_t slot, void *stream);
int rt_comm_gather_packed(rt_handle h, rt_comm_handle c, uint32_t stripe_rows, uint32_t slot, void *dst_image, void *stream);
int rt_comm_stage_accum(rt_handle h, rt_comm_handle c, uint32_t slot, void *stream);
/* Sums staging buffer `slot` over the ranks in place, alpha to 1; *image_out
 * receives its device address (width * height float4). */
int rt_comm_reduce_staged(rt_handle h, rt_comm_handle c, uint32_t slot, void **image_out, void *stream);

/* The two kernels of the stripe split on their own, for hosts that move the
 * bytes themselves (MPI, their own ncclSend/ncclRecv schedule): the first packs
 * the stripes rank, rank + world, ... of h's accumulator into `send`
 * (rt_stripes_packed_bytes() bytes of device memory: ceil(stripes / world)
 * stripes of stripe_rows rows, zero where the image ends), the second writes the
 * complete image to dst_image from `recv`, the ranks' packed buffers one after
 * the other in rank order.  Errors are reported through rt_comm_last_error(NULL). */
size_t rt_stripes_packed_bytes(uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world);
int rt_pack_stripes_to(rt_handle h, uint32_t stripe_rows, uint32_t world, uint32_t rank, void *send, void *stream);
int rt_unpack_stripes_from(rt_handle h, uint32_t stripe_rows, uint32_t world, const void *recv, void *dst_image,
                           void *stream);

/* Text of the last error on this communicator object; c may be NULL: then the
 * last failure of rt_comm_unique_id / rt_comm_create / rt_comm_adopt on this
 * thread. */
const char *rt_comm_last_error(rt_comm_handle c);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_RCCL_H */
