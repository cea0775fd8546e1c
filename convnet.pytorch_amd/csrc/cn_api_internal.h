// cn_api_internal.h -- internal declarations shared by the translation units of libconvnet_hip.so.
#pragma once
#include "cn_common.h"
#include <string.h>
