// fit_agg_hip.hpp — registration of the three aggregates over the GPU-resident state (fit_agg_hip.cpp).  The extension
// entry point (src/anofox_statistics_extension.cpp in the reference) calls these instead of
// RegisterOlsAggregateFunction / RegisterRidgeAggregateFunction / RegisterWlsAggregateFunction
// (src/aggregate_functions/ols_aggregate.cpp:377-426, ridge_aggregate.cpp:388-440, wls_aggregate.cpp:401-452).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "anofox_stats_hip.h"

namespace duckdb {
class ExtensionLoader;
struct FunctionData;
// diagnostics of a query's device states behind a bind data object of these aggregates (tests, EXPLAIN ANALYZE hooks):
// sums over the feature counts that occurred (normally one)
struct HipAggStats {
	AnofoxHipBatchOptions options; // what bind made of the options argument
	size_t widths;                 // distinct feature counts seen
	uint64_t rows_accepted;
	int64_t unrefined;             // groups returned as NULL with status 101
	uint64_t slot_high_water, live_slots, fit_calls, slots_fitted;
};
bool HipAggStatsOf(FunctionData &bind_data, HipAggStats &out); // false: not a bind data object of these aggregates
const void *HipAggSharedStateOf(FunctionData &bind_data);       // identity of the device states that Copy() shares
void RegisterHipOlsAggregateFunction(ExtensionLoader &loader);   // anofox_stats_ols_fit_agg, ols_fit_agg
void RegisterHipRidgeAggregateFunction(ExtensionLoader &loader); // anofox_stats_ridge_fit_agg, ridge_fit_agg
void RegisterHipWlsAggregateFunction(ExtensionLoader &loader);   // anofox_stats_wls_fit_agg, wls_fit_agg
} // namespace duckdb
