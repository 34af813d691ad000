#pragma once
#include "dump.h"
namespace opusfe { inline std::string dump_ir(const opusfe_unit*) { return ""; } }
