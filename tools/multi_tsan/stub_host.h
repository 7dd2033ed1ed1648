// stub_host.h - the context object of the host stand-ins (stub_scores.cpp, stub_search.cpp).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <string>

struct vsc_ctx {
    std::string err;
};
