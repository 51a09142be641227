// Resolved INSTEAD of the real <hip/hip_runtime.h> when the harness is compiled with -I tests/hip_emul first.
#pragma once
#include "../hip_emul.h"
