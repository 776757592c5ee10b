// gcsa.hpp -- everything the C++ layer over the C ABI (gcsa2_hip.h) offers, in one include.
//
// jltsiren/gcsa2 has no plugin ABI: its boundary is the class API of `gcsa::GCSA` and `gcsa::LCPArray`
// (reference include/gcsa/gcsa.h:40-277, include/gcsa/lcp.h:90-194).  That API lives in headers with the
// reference's own names and include paths -- <gcsa/utils.h>, <gcsa/files.h>, <gcsa/support.h>, <gcsa/gcsa.h>,
// <gcsa/lcp.h>, <gcsa/algorithms.h> under include/ -- so that existing callers compile unchanged; this file
// pulls them all in for code written against the engine directly.
#ifndef GCSA2_HIP_GCSA_HPP
#define GCSA2_HIP_GCSA_HPP

#include "../gcsa/utils.h"
#include "../gcsa/files.h"
#include "../gcsa/support.h"
#include "../gcsa/gcsa.h"
#include "../gcsa/lcp.h"
#include "../gcsa/algorithms.h"

#endif // GCSA2_HIP_GCSA_HPP
