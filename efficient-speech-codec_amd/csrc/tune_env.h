#pragma once
#include <cstdlib>
// Tuning switches vs product switches.  The default library reads only the ~20 PRODUCT switches through getenv (fallbacks the tests
// and tools/ab.py exercise, stream / part counts).  Every tuning switch of a measured-and-rejected kernel form goes through ESCX_TUNE_ENV, which is a
// constant null in the default build: the switch, its dead branch and - behind `#ifdef ESCX_EXPERIMENTAL` at the instantiation sites - the kernel
// forms themselves are not in libescx.so.  `ESCX_BUILD_TAG=exp ESCX_EXTRA_CXXFLAGS=-DESCX_EXPERIMENTAL python build.py` builds libescx_exp.so with all
// of them (ESCX_LIB_TAG=exp loads it); their A/B records are in profiles/ (r3_*, r4_*_ab.txt).
#ifndef ESCX_TUNE_ENV
#ifdef ESCX_EXPERIMENTAL
#define ESCX_TUNE_ENV(name) getenv(name)
#else
#define ESCX_TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif
#endif
