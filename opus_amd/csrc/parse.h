#pragma once
#include "unit.h"
namespace opusfe { inline void run_parser(opusfe_ctx* ctx, opusfe_unit* u) { u->status = OPUSFE_ERR_INTERNAL; ctx->msg = "parser not built yet"; } }
