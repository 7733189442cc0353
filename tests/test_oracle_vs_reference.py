"""CPU, build container only: the oracle against the GENUINE reference (imported from /root/reference, read-only) on the
options the committed fixtures do not cover - the shapes tests/test_gpu_parity.py::test_options_outside_the_shipped_configurations_...
and test_other_caption_lengths_... then hold the HIP path to.  Skipped wherever the reference tree is absent (the GPU box)."""
import pytest
import torch

from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_feats, synth_state_dict
from oracle import care_cpu
from oracle.ref_import import import_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="the reference tree is not on this machine")

@pytest.fixture(scope="module", autouse=True)
def _leave_no_trace():
    """The reference is imported with stand-in modules for two packages this image lacks (oracle/ref_import.py) and puts its
    tree on sys.path: both go again with this module, so that no later test sees them."""
    import sys

    modules, path = set(sys.modules), list(sys.path)
    yield
    for name in set(sys.modules) - modules:
        del sys.modules[name]
    sys.path[:] = path


VARIANTS = [
    ("frames8", "msrvtt_care", dict(n_frames=8)),
    ("vocab2003", "msrvtt_care", dict(vocab_size=2003)),
    ("alpha07", "msrvtt_care", dict(beam_alpha=0.7)),
    ("topk12_k300", "msrvtt_care", dict(use_attr_topk=12, attribute_prediction_k=300)),
    ("d256", "msrvtt_base_ami", dict(dim_hidden=256, num_attention_heads=4, intermediate_size=1024)),
    ("layers2", "msrvtt_care", dict(num_hidden_layers_decoder=2)),
    ("max_len12", "msrvtt_care", dict(max_len=12)),
    ("max_len45", "msrvtt_base_ami", dict(max_len=45)),
    ("sinusoid_pe", "msrvtt_care", dict(trainable_pe=False)),
    ("no_qkv_bias", "msrvtt_base_ami", dict(mha_exclude_bias=True)),
    ("no_hybrid_bias", "msrvtt_care", dict(add_hybrid_attention_bias=False)),
    ("decoder_mi", "msrvtt_base_ami", dict(modality_for_decoder="mi")),
    ("predictor_mi", "msrvtt_care", dict(modality_for_predictor="mi")),
    ("eps1e-6", "msrvtt_base_ami", dict(layer_norm_eps=1e-6)),
    ("modality_ai", "msrvtt_base_ami", dict(modality="ai")),
    ("share_prj", "msrvtt_care", dict(attribute_prediction_share_prj=True)),
    ("retrieval10", "msrvtt_care", dict(retrieval_topk=10)),
    ("dims_64_1024_768", "msrvtt_base_ami", dict(dim_a=64, dim_m=1024, dim_i=768)),
    ("dim_i_500", "msvd_base_i", dict(dim_i=500)),
    ("vocab100", "msrvtt_base_ami", dict(vocab_size=100)),
    ("d192", "msrvtt_care", dict(dim_hidden=192, num_attention_heads=3, intermediate_size=768)),
    ("d320_ff1280", "msrvtt_base_ami", dict(dim_hidden=320, num_attention_heads=5, intermediate_size=1280)),
]


@pytest.mark.parametrize("name,config,over", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_oracle_equals_the_reference_on_option_variants(name, config, over):
    get_framework, get_translator = import_reference()
    opt = make_opt(config, beam_size=5, topk=2, **over)
    torch.manual_seed(0)
    ref = get_framework(opt).eval()
    shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    sd = synth_state_dict(7, shapes, row_scale={"cls_head.tgt_word_prj.weight": {3: 4.0, 0: 3.0}})
    ref.load_state_dict(sd, strict=True)
    feats = synth_feats(7, feat_shapes(opt, 2))
    with torch.no_grad():
        r_hyps, r_scores = get_translator(opt).translate_batch([ref], {"feats": [f.clone() for f in feats]})
    hyps, scores = care_cpu.translate_batch(sd, opt, feats)
    assert hyps == r_hyps
    assert max(abs(a - b) for x, y in zip(scores, r_scores) for a, b in zip(x, y)) < 1e-5
    # and the module this repository builds for the same options has the reference's parameters, name by name
    from care_amd import get_framework as build
    assert [(k, tuple(v.shape)) for k, v in build(opt).state_dict().items()] == shapes


def test_prefix_guidance_is_a_layout_the_oracle_does_not_restate():
    """`use_attr_flags` Gp.. prepends the guidance vector to the decoder's input sequence (Embeddings.py:155-157); the oracle
    restates the additive form only and care_amd refuses the option (tests/test_host_cpu.py) - recorded here so that the gap is
    a known one: the reference and the oracle DO differ on it."""
    get_framework, get_translator = import_reference()
    opt = make_opt("msrvtt_care", beam_size=5, topk=1, use_attr_type="pp_emb_concat")
    torch.manual_seed(0)
    ref = get_framework(opt).eval()
    sd = synth_state_dict(7, [(k, tuple(v.shape)) for k, v in ref.state_dict().items()])
    ref.load_state_dict(sd, strict=True)
    feats = synth_feats(7, feat_shapes(opt, 2))
    with torch.no_grad():
        r_hyps, _ = get_translator(opt).translate_batch([ref], {"feats": [f.clone() for f in feats]})
    assert care_cpu.translate_batch(sd, opt, feats)[0] != r_hyps
