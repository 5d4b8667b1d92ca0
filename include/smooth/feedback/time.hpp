// Forwarding header: reference include path and namespace for the MI355X-native front
// (include/smooth_feedback_amd/time.hpp).  `smooth::feedback` aliases `smooth_feedback_amd`.
#pragma once
#include "../../smooth_feedback_amd/time.hpp"
namespace smooth { namespace feedback = ::smooth_feedback_amd; }
