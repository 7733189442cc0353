"""Detokenisation of hypotheses (`misc/utils.py:117-137` `to_sentence`)."""
from .constants import EOS, PAD


def to_sentence(hyp, vocab, break_words=(EOS, PAD), skip_words=(), extra_mappings=None, add_eos=False):
    table = {**vocab, **extra_mappings} if extra_mappings else vocab
    words, stop = [], False
    for wid in hyp:
        if stop:
            break
        if wid in skip_words:
            continue
        if wid in break_words:
            if add_eos and wid == EOS:
                stop = True
            else:
                break
        words.append(table[wid])
    return " ".join(words)
