#pragma once
#include "dump.h"
namespace opusfe { inline void ensure_host_ast(opusfe_unit*) {} inline std::string dump_ast(const opusfe_unit*) { return ""; } inline std::string dump_error_line(const opusfe_unit*) { return "Ok\n"; } }
