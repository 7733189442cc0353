"""Explicit `opt` dictionaries for the BASELINE.json configurations.

The reference builds its flat `opt` dict from argparse + YAML overlays
(`opts.py:15-354`, `config/{archs,tasks,methods,feats}.yaml`,
`models/Predictor/pred_attribute.py:168-210,308-340`).  That control plane is out of
scope; the hot path only ever reads `opt[...]`, so the keys it reads are spelled out
here (SURVEY.md section 8(b), row "opt keys read on the path").
"""
from copy import deepcopy

_ARCH = {
    # config/archs.yaml:1-26
    "base": dict(dim_hidden=512, num_attention_heads=8, intermediate_size=2048),
    "median": dict(dim_hidden=768, num_attention_heads=12, intermediate_size=3072),
    "large": dict(dim_hidden=1024, num_attention_heads=16, intermediate_size=4096),
}

_COMMON = dict(
    # config/methods.yaml:1-5 (Transformer) + opts.py:31-36 defaults
    encoder="Embedder",
    decoder="TransformerDecoder",
    cls_head="NaiveHead",
    decoding_type="ARFormer",
    fusion="temporal_concat",
    pointer=None,
    with_backbones=[],
    # config/archs.yaml:1-13
    encoder_dropout_prob=0.5,
    hidden_dropout_prob=0.5,
    attention_probs_dropout_prob=0.1,
    trainable_pe=True,
    hidden_act="relu",
    layer_norm_eps=1e-12,
    num_hidden_layers_decoder=1,
    num_hidden_layers_encoder=1,
    # opts.py:68-92,209-210
    transformer_pre_ln=False,
    mha_exclude_bias=False,
    enhance_input=2,
    with_category=False,
    RPE=False,
    n_frames=28,
    max_len=30,
    # notebooks/retrieval_robustness.ipynb:89 (MSRVTT); other corpora are not in the repo
    vocab_size=10547,
    # config/feats.yaml:1-9,46-51 (ViT = CLIP ViT-B/32 image feats, ResNeXt motion, VGGish audio)
    feats="ViT",
    dim_a=128,
    dim_m=2048,
    dim_i=512,
    crits=["lang"],
    # misc decode options (models/Translator.py:29-33)
    beam_size=1,
    beam_alpha=1.0,
    topk=1,
)

_CARE = dict(
    # config/tasks.yaml:7-52 (Concept -> CARE) + pred_attribute.py:168-210,308-340
    modality="amir",
    modality_for_decoder="ami",
    modality_for_predictor="amir",
    dim_r=512,
    retrieval_topk=20,
    crits=["lang", "attribute"],
    attribute_prediction=True,
    attribute_prediction_k=500,
    attribute_prediction_flags="V",
    attribute_prediction_mean_pooling=True,
    attribute_prediction_channel_concat=True,
    attribute_prediction_share_prj=False,
    attribute_prediction_sparse_sampling=False,
    use_attr=True,
    use_attr_flags="G1Lc",
    use_attr_type="emb_concat",
    use_attr_topk=30,
    add_hybrid_attention_bias=True,
    attr_layer_pos="cross2attr",
    predictors_to_be_added=["SemanticContainer"],
)


def make_opt(name: str, **overrides) -> dict:
    """Return the `opt` dict of one named configuration.

    Names (BASELINE.json `configs`, in order): ``msvd_base_i``, ``msrvtt_base_ami``,
    ``msrvtt_care``, ``vatex_care_large``, ``msrvtt_care_beam5``; plus variants used by
    the parity tests: ``care_median_gelu`` (archs.yaml:21-26 with GELU) and
    ``base_ami_mte`` (the working self-attention encoder, Encoder.py:190-193) and ``msrvtt_cabase``
    (the `attr_attention` local-guidance variant, Layers.py:117-119,139-154,218-225).
    """
    opt = deepcopy(_COMMON)
    if name == "msvd_base_i":
        opt.update(_ARCH["base"], modality="i")
    elif name == "msrvtt_base_ami":
        opt.update(_ARCH["base"], modality="ami")
    elif name == "msrvtt_care":
        opt.update(_ARCH["base"], **deepcopy(_CARE))
    elif name == "vatex_care_large":
        opt.update(_ARCH["large"], **deepcopy(_CARE))
    elif name == "msrvtt_care_beam5":
        opt.update(_ARCH["base"], **deepcopy(_CARE))
        opt.update(beam_size=5)
    elif name == "msrvtt_care_g1l0":
        # scripts/exp_ablation_main.sh:34,63 `--use_attr_flags G1L0` (no --add_hybrid_attention_bias): global guidance only -
        # check_args maps the flags to use_attr_type "emb_" (pred_attribute.py:308-330); no concept rows in the memory
        opt.update(_ARCH["base"], **deepcopy(_CARE))
        opt.update(use_attr_flags="G1L0", use_attr_type="emb_", add_hybrid_attention_bias=False)
    elif name == "msrvtt_care_g0l0":
        # scripts/exp_ablation_main.sh:37,66 `--use_attr_flags G0L0`: check_args turns use_attr off (pred_attribute.py:309-310):
        # the concept head is trained and predicted (crits lang + attribute), the decoder gets no guidance at all
        opt.update(_ARCH["base"], **deepcopy(_CARE))
        opt.update(use_attr=False, use_attr_flags="G0L0", use_attr_type="", add_hybrid_attention_bias=False,
                   predictors_to_be_added=[])
    elif name == "care_median_gelu":
        opt.update(_ARCH["median"], **deepcopy(_CARE))
        opt.update(hidden_act="gelu")
    elif name == "msrvtt_cabase":
        # config/tasks.yaml:56-61 (CABase): no global guidance, local guidance by a third attention
        # over the concept embeddings ("Cross -> Semantic"), visual-driven concept detection, no bias
        opt.update(_ARCH["base"], **deepcopy(_CARE))
        opt.update(modality="ami", modality_for_decoder="ami", modality_for_predictor="mi",
                   use_attr_flags="G0L1", use_attr_type="_att", attr_layer_pos="cross2attr",
                   add_hybrid_attention_bias=False)
        opt.pop("dim_r", None)
    elif name == "base_ami_mte":
        opt.update(_ARCH["base"], modality="ami", encoder="MultiTransformerEncoder")
    else:
        raise ValueError("unknown configuration `{}`".format(name))
    opt.update(overrides)
    return opt


CONFIG_NAMES = (
    "msvd_base_i",
    "msrvtt_base_ami",
    "msrvtt_care",
    "vatex_care_large",
    "msrvtt_care_beam5",
    "care_median_gelu",
    "base_ami_mte",
    "msrvtt_cabase",
    "msrvtt_care_g1l0",
    "msrvtt_care_g0l0",
)


def feat_shapes(opt: dict, batch: int):
    """Shapes of `batch['feats']` in modality order (dataloader layout)."""
    shapes = []
    for ch in opt["modality"]:
        n = opt["retrieval_topk"] if ch == "r" else opt["n_frames"]
        shapes.append((batch, n, opt["dim_" + ch]))
    return shapes
