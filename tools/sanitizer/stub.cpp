// gl_fail / last-error storage normally provided by ntt.hip (which needs the HIP runtime); the harness links only verifier.hip
#include <string>
thread_local std::string g_gl_last_error;
int gl_fail(int code, const char* what, const char* file, int line) { g_gl_last_error = what; (void)file; (void)line; return code; }
