// version.hip - what a built library says about itself.  care_amd/build.py compiles this file on every link with
//   -DCARE_SRC_HASH_STR="<sha-256 of csrc/*.hip, csrc/*.h, include/care_hip.h and the compile flags>"
//   -DCARE_BUILD_FLAGS_STR="<the variant's / tool's extra flags>"
// and care_amd/_lib.py compares care_source_hash() with the hash of the tree it runs from: a library that does not
// come from these sources is rebuilt or refused, never silently used.
#include "care_common.h"

#ifndef CARE_SRC_HASH_STR
#define CARE_SRC_HASH_STR "unhashed------------------------"
#endif
#ifndef CARE_BUILD_FLAGS_STR
#define CARE_BUILD_FLAGS_STR ""
#endif

// (the marker in front lets build.py read the hash out of the file without loading it)
static const char care_hash_blob[] = "CARE_SRC_HASH=" CARE_SRC_HASH_STR;

extern "C" int care_version(void) { return CARE_ABI_VERSION; }
extern "C" const char* care_arch(void) { return "gfx950"; }
extern "C" const char* care_source_hash(void) { return care_hash_blob + 14; }
extern "C" const char* care_build_flags(void) { return CARE_BUILD_FLAGS_STR; }
// the 16-bit storage / MFMA operand type this library was compiled for: "bf16" or "fp16" (-DCARE_H16_FP16)
extern "C" const char* care_h16(void) { return CARE_H16_NAME; }
