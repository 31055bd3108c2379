// One translation unit per backward-kernel variant of the training path: -DGBNF_V_ARGS="KIND,HT,OT,ACTA,ACTB,DEPTH"
#include "gbnf_train_bwd.hip.h"
#ifndef GBNF_V_ARGS
#error "compile with -DGBNF_V_ARGS=KIND,HT,OT,ACTA,ACTB,DEPTH"
#endif
#define GBNF_INST2(...) GBNF_INSTANTIATE_HX3_BWD(__VA_ARGS__)
GBNF_INST2(GBNF_V_ARGS)
