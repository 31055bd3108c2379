// One translation unit per hx3 kernel variant: -DGBNF_V_ARGS="KIND,HT,OT,NT,ACTA,ACTB,PREC,DEPTH" [-DGBNF_V_TRAIN=1]
#include "gbnf_flow_kernel_hx3.hip.h"
#ifndef GBNF_V_ARGS
#error "compile with -DGBNF_V_ARGS=KIND,HT,OT,NT,ACTA,ACTB,PREC,DEPTH"
#endif
#ifndef GBNF_V_TRAIN
#define GBNF_V_TRAIN 0
#endif
#define GBNF_INST2(...) GBNF_INSTANTIATE_HX3_T(__VA_ARGS__)
GBNF_INST2(GBNF_V_ARGS, GBNF_V_TRAIN)
