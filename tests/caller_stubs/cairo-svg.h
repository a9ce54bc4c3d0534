#pragma once
#include "cairo.h"
