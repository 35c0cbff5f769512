/*
 * transoar_criterion.h -- C ABI of the fused set criterion of the training step for MI355X (gfx950).
 *
 * Reference semantics: transoar/models/criterion.py:9-125 (per-class matching, BCE on soft labels, L1 + GIoU on the
 * matched boxes, the auxiliary-loss quirk of :114-123: every intermediate decoder layer is matched on ITS logits, but
 * the box and class losses are taken on the FINAL outputs) with the matcher of transoar/models/matcher.py:9-65 in its
 * anchor-matching form (cost = cost_class * -sigmoid(logit) + cost_giou * -GIoU(anchor, target)
 * + cost_bbox * L1(anchor, target), the cheapest query of a class is its match, soft labels = min-max normalised GIoU of
 * the class's anchors against the target, -1 for classes without a target) and the box helpers of
 * transoar/utils/bboxes.py:6-43.  The host-side mirror (transoar_amd/criterion.py, matcher.py, bboxes.py) computes the
 * same with ~270 five-microsecond torch launches per step; this is one launch forward and one backward.
 *
 * Shapes: N samples, O organ classes, R queries per class (Q = O * R queries, query = class-major), LAYERS decoder
 * outputs (entry 0 = the FINAL layer, entries 1.. = the auxiliary ones in their order).
 * All pointers are device pointers; asynchronous on `hip_stream`; capturable (no host synchronisation, the
 * normalisers may live in device memory).  Returns 0, a hipError_t (> 0), or a negative TRANSOAR_CRIT_ERR_* code.
 */
#ifndef TRANSOAR_CRITERION_H
#define TRANSOAR_CRITERION_H
#ifdef __cplusplus
extern "C" {
#endif

#define TRANSOAR_CRIT_MAX_LAYERS 8
#define TRANSOAR_CRIT_MAX_R 64
#define TRANSOAR_CRIT_ERR_NULL (-1)
#define TRANSOAR_CRIT_ERR_DIM (-2)
#define TRANSOAR_CRIT_ERR_DTYPE (-3)

/* element types of logits / boxes */
#define TRANSOAR_CRIT_F32 0
#define TRANSOAR_CRIT_BF16 2

/*
 * Forward.
 *   logits[l]       (N, Q) class logits of decoder output l (one logit per query), dtype logits_dtype
 *   boxes           (N, Q, 6) predicted boxes (cx, cy, cz, w, h, d) of the FINAL layer, dtype boxes_dtype
 *   anchors         (Q, 6) fp32;  tgt_boxes (N, O, 6) fp32;  present (N, O) bytes, non-zero = the class has a target
 *   num_boxes_dev   device float, or NULL: then num_boxes_host is the normaliser of the box losses (criterion.py:96)
 *   n_present_dev   device float = number of (sample, class) slots with a target summed over the data-parallel ranks, or
 *                   NULL: then the class loss is the mean over this batch's valid entries (criterion.py:46-49)
 *   losses          (5 + 3 * (layers - 1)) fp32 out, in the order of the criterion's dict (criterion.py:98-124):
 *                   [bbox, giou, cls, segce = 0, segdice = 0] of the final output, then [bbox_i, giou_i, cls_i] of every
 *                   auxiliary output (cls_i = cls: with anchor matching the soft labels do not depend on the output)
 *   d_l1, d_giou    (N, Q, 6) fp32 out: d(L1) / d(box) and d(1 - GIoU) / d(box) of every query, divided by num_boxes
 *   d_cls           (N, Q) fp32 out: d(class loss) / d(final logit)
 *   hit             (layers, N, Q) bytes out: 1 where the query is its class's match on that layer
 */
int transoar_set_criterion_forward(const void* const* logits, int layers, int logits_dtype, const void* boxes, int boxes_dtype,
                                   const float* anchors, const float* tgt_boxes, const unsigned char* present,
                                   const float* num_boxes_dev, float num_boxes_host, const float* n_present_dev,
                                   float cost_class, float cost_bbox, float cost_giou, int N, int O, int R,
                                   float* losses, float* d_l1, float* d_giou, float* d_cls, unsigned char* hit, void* hip_stream);

/*
 * Backward: grad_boxes (N, Q, 6) = a * d_l1 + b * d_giou with a = sum_l g[bbox_l] * hit_l, b = sum_l g[giou_l] * hit_l (zeros for
 *           a query no layer matched: like torch.where in criterion.py:68-74, never 0 * NaN), dtype boxes_dtype;
 *           grad_logits (N, Q) = (sum_l g[cls_l]) * d_cls, dtype logits_dtype (the FINAL layer's logits: the auxiliary logits only
 *           steer the matching and receive no gradient);   g = gradient of the `losses` vector (device, fp32, same layout).
 */
int transoar_set_criterion_backward(const float* g, int layers, const float* d_l1, const float* d_giou, const float* d_cls,
                                    const unsigned char* hit, int N, int O, int R, void* grad_boxes, int boxes_dtype,
                                    void* grad_logits, int logits_dtype, void* hip_stream);

int transoar_criterion_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
