// stand-in: see ../duckdb.hpp
#pragma once
#include "duckdb.hpp"
