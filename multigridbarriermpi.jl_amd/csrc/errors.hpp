// Typed exceptions of the library; the C ABI firewall (capi.cpp: guard) maps the TYPE to a status code of
// include/mgb_hip.h -- never the message text.
#pragma once
#include <stdexcept>
#include <string>

namespace mgb {

struct ArgError : std::invalid_argument {      // MGB_E_ARG: bad argument / unknown name / shape mismatch
  using std::invalid_argument::invalid_argument;
};
struct HipError : std::runtime_error {         // MGB_E_HIP: HIP runtime failure, or no GPU visible
  using std::runtime_error::runtime_error;
};
struct NumericError : std::runtime_error {     // MGB_E_NUMERIC: non-SPD Hessian, infeasible start, kappa collapse
  using std::runtime_error::runtime_error;
};
struct InternalError : std::logic_error {      // MGB_E_INTERNAL: a broken invariant of the library itself
  using std::logic_error::logic_error;
};

}  // namespace mgb
