function [Z, support] = mmv_omp(A, K, Y)
% Joint (MMV) OMP: replaces  s = spx.pursuit.joint.OrthogonalMatchingPursuit(A, K); r = s.solve(Y); r.Z
% of the drivers (plot_errorVSsnr.m:116-117) without the sparse-plex toolbox.
  [Z, support] = jstsp_mex('mmv_omp', A, K, Y);
end
