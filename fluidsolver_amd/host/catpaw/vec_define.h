// Host array aliases of the plugin surface (catpaw/vec_define.h:18-21 in the reference).
#pragma once

#include <vector>

#include "geometry.h"

using std::vector;

typedef std::vector<float> vecf;
typedef std::vector<int> veci;
typedef std::vector<cfloat3> vecf3;
typedef std::vector<cfloat4> vecf4;
typedef std::vector<adjacent> vecadj;
