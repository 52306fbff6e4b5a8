// main.cpp of the reference includes this header with a Windows path separator
// (particle_solver/main.cpp:3); on a POSIX file system that is a file of this very name.
#pragma once
#include "catpaw/geometry.h"
