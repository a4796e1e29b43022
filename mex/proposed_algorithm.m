function [S, Y, convergence_error] = proposed_algorithm(subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type)
% Drop-in for basic_system_functions/proposed_algorithm.m (same signature); runs on the MI355X
% through libjstsp_mi355x.so.  Put this directory ahead of the reference's on the MATLAB path.
  if nargout >= 3
    [S, Y, convergence_error] = jstsp_mex('proposed_algorithm', subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type);
  elseif nargout == 2
    [S, Y] = jstsp_mex('proposed_algorithm', subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type);
  else
    S = jstsp_mex('proposed_algorithm', subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type);
  end
end
