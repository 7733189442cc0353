// decode_resident_beam_wide.hip - the resident beam launch for beam sizes 6 .. 8 (opts.py --beam_size; the reference's default is 5):
// decode_resident_beam.hip compiled once more with 8 groups kept per (row, vocabulary part) instead of 5 - the vocabulary phase's
// per-lane lists, the advance phase's merge and candidate tiles are written in RES_BMK (csrc/decode_resident.h) - under the names
// care_decode_resident_beam8 / care_decode_resident_beam8_scratch, which care_decode_resident_beam / _scratch route beam > 5 to.
// A second instance instead of a wider default: the lists live in registers through the vocabulary phase's epilogue, and beam 5 -
// translate.py's default - keeps the instance sized for it.
#define CARE_RES_BMK 8
#include "decode_resident_beam.hip"
