// checker.hip — placeholder until the device checkers land.
#include "engine_internal.h"
int msim_check_launch(msim_ctx *ctx) { ctx->err = "checker not built yet"; return MSIM_E_UNSUPPORTED; }
