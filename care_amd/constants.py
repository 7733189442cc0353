"""Token-id constants of the captioning vocabulary.

Same numbering as the reference (`config/Constants.py:1-6`): the decoder masks PAD
keys, beams start at BOS and stop at EOS.
"""
PAD = 0
UNK = 1
BOS = 2
EOS = 3
MASK = 4
VIS = 5
