"""Key maps from the ORIGINAL pretrained checkpoints to the VOLTA state_dict this package (and the reference) loads.

* UC2: volta/conversions/convert_uc2.py:29-68 -- rename rules from the UC2 authors' ``model_step_200000.pt`` to
  VOLTA's ``BertForVLPreTraining`` keys; keys that have no VOLTA counterpart (mask embedding, visual heads) are
  omitted, like the reference.
* M3P: volta/conversions/M3P_volta.ipynb -- ``module.X`` -> ``bert.encoder.X`` on ``checkpoint['model']``.

Both return plain dicts of tensors (no pickled objects): save them with ``torch.save`` and load with
``BertForVLTasks.from_pretrained`` / ``M3PForVLTasks.from_pretrained``.
"""


def uc2_original_to_volta_key(key):
    """One key of the original UC2 checkpoint -> its VOLTA name (convert_uc2.py:33-54)."""
    ln = str(key).replace("roberta", "bert")
    for a, b in (("img_embeddings", "embeddings"), ("img_linear", "image_embeddings"),
                 ("pos_linear", "image_location_embeddings"), ("img_layer_norm", "image_layer_norm"),
                 ("pos_layer_norm", "image_location_layer_norm"), ("attention.self", "attention_self"),
                 ("attention.output", "attention_output")):
        ln = ln.replace(a, b)
    if ".layer." in ln:  # BERT layer n -> VOLTA sub-layers 2n (attention) and 2n+1 (feed-forward)
        num = int(ln.split(".")[3])
        new = 2 * num + int(".intermediate." in ln or ".output." in ln)
        ln = ln.replace(".%d." % num, ".%d." % new)
    for a, b in (("pooler", "t_pooler"), ("cls.dense", "cls.predictions.transform.dense"),
                 ("cls.layer_norm", "cls.predictions.transform.LayerNorm"), ("cls.bias", "cls.predictions.bias"),
                 ("cls.decoder", "cls.predictions.decoder"), ("itm_output", "cls.bi_seq_relationship")):
        ln = ln.replace(a, b)
    return ln


def convert_uc2(original_state_dict, target_state_dict):
    """Returns (new_state_dict, omitted_keys): ``target_state_dict`` with every tensor that has a counterpart in the
    original checkpoint replaced by it (shapes must agree, convert_uc2.py:63)."""
    out = dict(target_state_dict)
    omitted = []
    for k, v in original_state_dict.items():
        ln = uc2_original_to_volta_key(k)
        if ln not in out:
            omitted.append(k)
            continue
        if tuple(out[ln].shape) != tuple(v.shape):
            raise ValueError("convert_uc2: shape mismatch for %s <- %s: %s vs %s" % (ln, k, tuple(out[ln].shape), tuple(v.shape)))
        out[ln] = v
    return out, omitted


def convert_m3p(original_model_dict):
    """``checkpoint['model']`` of the M3P authors' file -> VOLTA keys (M3P_volta.ipynb)."""
    return {k.replace("module.", "bert.encoder."): v for k, v in original_model_dict.items()}
