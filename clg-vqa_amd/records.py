"""Record -> batch on the host: the input side of the hot path.

Restates, vectorised over the whole batch, what the reference does per sample in Python:

* ``BertPreprocessBatch.__call__`` (volta/volta/datasets/gqa_dataset_semantic_code_mix.py:564-651): box-location
  features from pixel boxes -- area first, on the RAW coordinates ``(y2-y1)(x2-x1)/(w h)`` (:586-591), then
  ``x/w, y/h`` (:594-597), then for ``num_locs > 5`` width and height of the NORMALISED box in columns 4, 5 (:599-601)
  => UC2 7-d ``[x1,y1,x2,y2,w,h,area]``, M3P 5-d ``[x1,y1,x2,y2,area]``; with ``norm_embeddings`` (M3P) the feature
  rows are L2-normalised with ``F.normalize`` (eps 1e-12) and the location rows divided by their plain L2 norm (:603-606:
  an all-zero padded row becomes NaN there, Appendix B quirk 4 -- reproduced, the 100-box extractor never pads);
  token truncation ``[t0] + t[1:-1][:seq_len-2] + [t-1]`` and padding with ``padding_index`` (:622-623, :691-700).
* ``GQAClassificationLoader.__iter__`` (:386-452): ``target.scatter_(1, labels, scores)``, the 10-tuple.
* ``get_embeddingdist`` (:371-381): the ``[B, num_labels]`` prior-distance rows, a 1842-iteration Python loop per
  sample over a ``(j, t)``-keyed dict in the reference; here ONE row gather from a dense ``[C, C]`` table built once
  (``prior_table``).

The tokenizer (question string -> ids) and the LMDB / tensorpack reader stay outside: a record carries ``tokens``.
Batches come out as CPU tensors in the reference's 10-tuple layout, ready for ``clg_vqa_amd.data.DevicePrefetcher``.
"""
import numpy as np
import torch


def prior_table(semantic_dict, num_labels):
    """Dense table T with T[t, j] = semantic_dict[(j, t)] for j != t and 0 on the diagonal, so that the reference's
    ``get_embeddingdist(labels)[i] == T[labels[i][-1]]`` (gqa_dataset_semantic_code_mix.py:371-381).  fp64 like the
    reference's ``np.zeros(..., dtype=float)`` (the loader casts the rows to fp32 afterwards)."""
    T = np.zeros((num_labels, num_labels), dtype=np.float64)
    for (j, t), v in semantic_dict.items():
        if j != t:
            T[t, j] = v
    return T


def box_locations(boxes, img_w, img_h, num_locs):
    """boxes [..., 4] pixel (x1, y1, x2, y2) (zero rows = padding), img_w / img_h broadcastable to boxes[..., 0]
    -> [..., num_locs] float32, the reference's arithmetic in float32 and in its order of operations."""
    boxes = np.asarray(boxes, dtype=np.float32)
    loc = np.zeros(boxes.shape[:-1] + (num_locs,), dtype=np.float32)
    loc[..., :4] = boxes
    # the reference divides float32 arrays by python floats: the scalar (w, h, or the DOUBLE product w*h) is rounded
    # to float32 and the division itself is a float32 one
    w64 = np.asarray(img_w, dtype=np.float64)
    h64 = np.asarray(img_h, dtype=np.float64)
    w, h, wh = w64.astype(np.float32), h64.astype(np.float32), (w64 * h64).astype(np.float32)
    if num_locs >= 5:
        loc[..., -1] = (loc[..., 3] - loc[..., 1]) * (loc[..., 2] - loc[..., 0]) / wh
    loc[..., 0] = loc[..., 0] / w
    loc[..., 1] = loc[..., 1] / h
    loc[..., 2] = loc[..., 2] / w
    loc[..., 3] = loc[..., 3] / h
    if num_locs > 5:
        loc[..., 4] = loc[..., 2] - loc[..., 0]
        loc[..., 5] = loc[..., 3] - loc[..., 1]
    return loc


def collate_records(records, seq_len, region_len, num_locs, num_labels, padding_index=1, norm_embeddings=False,
                    prior=None, batch_index=0, feat_dim=2048):
    """records: dicts with ``features`` [n, feat_dim] f32, ``boxes`` [n, 4] f32 (pixels), ``img_w``, ``img_h``,
    ``tokens`` (ids incl. <s> ... </s>), ``labels`` [L] int, ``scores`` [L] float, ``question_id``.
    prior: the dense table of ``prior_table`` (or None: zeros).  Returns the reference's 10-tuple of CPU tensors."""
    B = len(records)
    feats = np.zeros((B, region_len, feat_dim), dtype=np.float32)
    boxes = np.zeros((B, region_len, 4), dtype=np.float32)
    image_mask = np.zeros((B, region_len), dtype=np.int64)
    ids = np.full((B, seq_len), padding_index, dtype=np.int64)
    input_mask = np.zeros((B, seq_len), dtype=np.int64)
    w = np.empty((B, 1), dtype=np.float64)
    h = np.empty((B, 1), dtype=np.float64)
    for b, r in enumerate(records):  # ragged parts only: copies, no arithmetic
        n = len(r["boxes"])
        f = np.asarray(r["features"], dtype=np.float32).reshape(-1, feat_dim)
        if norm_embeddings:
            f = torch.nn.functional.normalize(torch.from_numpy(f.copy()), dim=-1).numpy()
        feats[b, :n] = f
        boxes[b, :n] = np.asarray(r["boxes"], dtype=np.float32).reshape(-1, 4)
        image_mask[b, :n] = 1
        t = list(r["tokens"])
        t = [t[0]] + t[1:-1][:seq_len - 2] + [t[-1]]
        ids[b, :len(t)] = t
        input_mask[b, :len(t)] = 1
        w[b, 0], h[b, 0] = float(r["img_w"]), float(r["img_h"])
    loc = box_locations(boxes, w, h, num_locs)
    if norm_embeddings:
        with np.errstate(invalid="ignore", divide="ignore"):
            loc = loc / np.linalg.norm(loc, 2, -1, keepdims=True)
    labels = np.stack([np.asarray(r["labels"], dtype=np.int64) for r in records])
    scores = np.stack([np.asarray(r["scores"], dtype=np.float32) for r in records])
    target = torch.zeros((B, num_labels), dtype=torch.float32)
    target.scatter_(1, torch.from_numpy(labels), torch.from_numpy(scores))
    if prior is not None:
        dist = torch.from_numpy(np.ascontiguousarray(prior[labels[:, -1]])).to(torch.float32)  # one row gather
    else:
        dist = torch.zeros((B, num_labels), dtype=torch.float32)
    qid = torch.tensor([int(r["question_id"]) for r in records])
    return (torch.from_numpy(feats), torch.from_numpy(loc.astype(np.float32)), torch.from_numpy(image_mask),
            torch.from_numpy(ids), target, torch.from_numpy(input_mask), torch.zeros((B, seq_len), dtype=torch.int64),
            qid, torch.tensor(batch_index), dist)
