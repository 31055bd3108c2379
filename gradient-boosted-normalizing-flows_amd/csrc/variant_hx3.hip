// One translation unit per hx3 kernel variant: -DGBNF_V_ARGS="KIND,HT,OT,NT,ACTA,ACTB,PREC,DEPTH"
#include "gbnf_flow_kernel_hx3.hip.h"
#ifndef GBNF_V_ARGS
#error "compile with -DGBNF_V_ARGS=KIND,HT,OT,NT,ACTA,ACTB,PREC,DEPTH"
#endif
#define GBNF_INST2(...) GBNF_INSTANTIATE_HX3(__VA_ARGS__)
GBNF_INST2(GBNF_V_ARGS)
