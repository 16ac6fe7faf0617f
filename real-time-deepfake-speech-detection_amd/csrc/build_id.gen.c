const char afx_build_id_str[] = "037c91e27bb3";
