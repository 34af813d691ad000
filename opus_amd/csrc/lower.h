#pragma once
#include "unit.h"
namespace opusfe { inline void run_lower(opusfe_ctx* ctx, opusfe_unit* u) { u->status = OPUSFE_ERR_INTERNAL; ctx->msg = "IR generation not built yet"; }
inline void free_ir(opusfe_ctx*, opusfe_unit*) {} }
