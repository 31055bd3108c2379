// One translation unit per kernel variant: compiled with
//   -DGBNF_V_ARGS="KIND,HT,KSL,KS1,OT,NT,LMID,ACTA,ACTB"      (see variants.list / build.py)
#include "gbnf_flow_kernel.hip.h"
#ifndef GBNF_V_ARGS
#error "compile with -DGBNF_V_ARGS=KIND,HT,KSL,KS1,OT,NT,LMID,ACTA,ACTB"
#endif
#define GBNF_INST2(...) GBNF_INSTANTIATE(__VA_ARGS__)
GBNF_INST2(GBNF_V_ARGS)
