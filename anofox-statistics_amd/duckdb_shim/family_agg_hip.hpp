// family_agg_hip.hpp — registration of the rest of the regression family over the batched C ABI (family_agg_hip.cpp).  The
// extension entry point (src/anofox_statistics_extension.cpp:184-199 in the reference) calls these instead of
//   RegisterOlsFitPredictAggregateFunction / Ridge.. / Wls..   (src/aggregate_functions/*_predict_aggregate.cpp)
//   RegisterOlsFitPredictFunction / Ridge.. / Wls..            (src/window_functions/*_fit_predict.cpp)
//   RegisterVifAggregateFunction                               (src/aggregate_functions/vif_aggregate.cpp:201-232)
#pragma once

namespace duckdb {
class ExtensionLoader;
void RegisterHipOlsFitPredictAggregateFunction(ExtensionLoader &loader);   // anofox_stats_ols_fit_predict_agg, ols_fit_predict_agg, ols_predict_agg, anofox_stats_ols_predict_agg
void RegisterHipRidgeFitPredictAggregateFunction(ExtensionLoader &loader); // .._ridge_..
void RegisterHipWlsFitPredictAggregateFunction(ExtensionLoader &loader);   // .._wls_..
void RegisterHipOlsFitPredictFunction(ExtensionLoader &loader);            // anofox_stats_ols_fit_predict, ols_fit_predict (window)
void RegisterHipRidgeFitPredictFunction(ExtensionLoader &loader);
void RegisterHipWlsFitPredictFunction(ExtensionLoader &loader);
void RegisterHipVifAggregateFunction(ExtensionLoader &loader);             // anofox_stats_vif_agg, vif_agg
} // namespace duckdb
