#pragma once
// radx.hpp — single include of the HIP-backed RadX host API (reference: include/radx/radx.hpp:1-10).
//
// Same class names, method names and call order as the reference's header-only API
// (include/radx/radx_internal.hpp, radx_device.hpp, radx_buffer.hpp); the Vulkan handle types
// are replaced by HIP-backed PODs (radx_core.hpp) and all device work goes through the C ABI
// in include/radx_hip.h.  Link against radx_amd/lib/libradx_hip.so and libamdhip64.
#include "radx_core.hpp"
#include "radx_utils.hpp"
#include "radx_device.hpp"
#include "radx_buffer.hpp"
#include "radx_internal.hpp"
