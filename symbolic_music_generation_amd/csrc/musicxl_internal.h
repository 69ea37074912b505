// Internal glue: pulls in the public C ABI so kernels and the header cannot drift apart.
#pragma once
#include "../../include/musicxl.h"
