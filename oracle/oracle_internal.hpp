// ORACLE — TEST INFRASTRUCTURE ONLY (see opus_oracle.hpp).
#pragma once
#include "opus_oracle.hpp"

namespace oracle {
int op_precedence(Op o);          // parse.rs:205-222
const char* op_symbol(Op o);      // parse.rs:224-243
void get_definitions(Unit& u);    // frontend.rs:120-138
void lex_only(Unit& u);           // lexer alone (KAT-L)
void generate_ir(Unit& u);        // frontend.rs:162-166 over generate_ir.rs:100-116
}  // namespace oracle
