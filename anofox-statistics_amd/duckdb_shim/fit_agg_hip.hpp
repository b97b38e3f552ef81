// fit_agg_hip.hpp — registration of the three aggregates over the GPU-resident state (fit_agg_hip.cpp).  The extension
// entry point (src/anofox_statistics_extension.cpp in the reference) calls these instead of
// RegisterOlsAggregateFunction / RegisterRidgeAggregateFunction / RegisterWlsAggregateFunction
// (src/aggregate_functions/ols_aggregate.cpp:377-426, ridge_aggregate.cpp:388-440, wls_aggregate.cpp:401-452).
#pragma once

namespace anofox_shim {
class AggArena;
}

namespace duckdb {
class ExtensionLoader;
struct FunctionData;
// the query's arena behind a bind data object of these aggregates (diagnostics and tests: rows accepted, groups
// flagged as unrefined, slots in use); nullptr for any other bind data
anofox_shim::AggArena *HipAggArenaOf(FunctionData &bind_data);
void RegisterHipOlsAggregateFunction(ExtensionLoader &loader);   // anofox_stats_ols_fit_agg, ols_fit_agg
void RegisterHipRidgeAggregateFunction(ExtensionLoader &loader); // anofox_stats_ridge_fit_agg, ridge_fit_agg
void RegisterHipWlsAggregateFunction(ExtensionLoader &loader);   // anofox_stats_wls_fit_agg, wls_fit_agg
} // namespace duckdb
