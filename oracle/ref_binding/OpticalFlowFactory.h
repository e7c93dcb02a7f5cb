// oracle/ref_binding/OpticalFlowFactory.h — TEST INFRASTRUCTURE. The "two extra factory lines" of INTEGRATION.md section 1
// without touching (or copying) the reference's SR/optical_flow/OpticalFlowFactory.h: this file is force-included in
// front of every source of the reference's program (-include), pulls in the reference's header with its factory
// function renamed, and defines the factory the callers see — the two *_hip names first, everything else delegated.
// The callers' own #include "OpticalFlowFactory.h" then finds the reference's header already included (#pragma once).
#pragma once
#define makeOpticalFlowByName makeOpticalFlowByName_reference
#include <optical_flow/OpticalFlowFactory.h>  // the reference's, by its path under -I$(REF)
#undef makeOpticalFlowByName
#include "PixFlowHip.h"

namespace surround360 {
namespace optical_flow {

static OpticalFlowInterface* makeOpticalFlowByName(const string flowAlgName) {
  if (flowAlgName == "pixflow_low_hip")       return new PixFlowHip(globalS360Ctx(), "pixflow_low");
  if (flowAlgName == "pixflow_search_20_hip") return new PixFlowHip(globalS360Ctx(), "pixflow_search_20");
  return makeOpticalFlowByName_reference(flowAlgName);
}

}  // namespace optical_flow
}  // namespace surround360
