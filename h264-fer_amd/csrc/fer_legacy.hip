// fer_legacy.hip -- placeholder translation unit for the legacy global-state seam
// (RBSP_encode / frame / NALunit); filled in by fer_legacy shims.
#include "fer_internal.h"
