// Console colour escapes used by the messages of the hot-path classes (the reference prints its warnings and errors
// to std::cout with these prefixes, e.g. CubeHandler.h:147-151, ICP.cpp:159-163).
#pragma once
#ifndef RESET
#define RESET "\033[0m"
#define RED "\033[31m"
#define GREEN "\033[32m"
#define YELLOW "\033[33m"
#define BLUE "\033[34m"
#endif
