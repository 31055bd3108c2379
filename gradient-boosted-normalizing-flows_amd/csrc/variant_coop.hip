// One translation unit per cooperative (latency-form) kernel variant: -DGBNF_V_ARGS="KIND,HT,OT,FORM,ACTA,ACTB" (FORM: gbnf_flow_kernel_coop.hip.h)
#include "gbnf_flow_kernel_coop.hip.h"
#ifndef GBNF_V_ARGS
#error "compile with -DGBNF_V_ARGS=KIND,HT,OT,FORM,ACTA,ACTB"
#endif
#define GBNF_INST2(...) GBNF_INSTANTIATE_COOP(__VA_ARGS__)
GBNF_INST2(GBNF_V_ARGS)
