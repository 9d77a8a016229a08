// Stand-in for <cuda_fp16.h>: everything lives in cuda_shim.h.
#pragma once
#include "cuda_shim.h"
